#!/usr/bin/env python3
"""The isolated roofline probe of bench.py as rocprofv3 saw it:  python tools/probe_from_trace.py <kernel_trace.csv> [kernel substring]

bench.py's probe launches the roofline layer 250 times back to back (50 warm-up + 200 timed, events on the launch stream); in the kernel
trace of the same command that is the longest run of consecutive launches of the kernel.  Prints the run's length and the average
duration of its last 200 launches (the timed ones) - the figure `roofline.avg_kernel_us` of the bench line has to agree with (the
bench figure includes the inter-launch gaps of back-to-back launches, the trace's durations do not)."""
import csv
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "conv_rows_ksplit<128, false, 64, 1, 0, false>"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
best, cur = [], []
for s, e, name in rows:
    if want in name:
        cur.append((s, e))
    else:
        if len(cur) > len(best):
            best = cur
        cur = []
if len(cur) > len(best):
    best = cur
if not best:
    sys.exit(f"no launch of {want}")
timed = best[-200:]
dur = [e - s for s, e in timed]
span = (timed[-1][1] - timed[0][0]) / len(timed)
print(f"{want}: longest back-to-back run {len(best)} launches; last {len(timed)}: kernel duration avg {sum(dur) / len(dur) / 1e3:.2f} us "
      f"(min {min(dur) / 1e3:.2f}, max {max(dur) / 1e3:.2f}), start-to-start {span / 1e3:.2f} us")
