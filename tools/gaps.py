#!/usr/bin/env python3
"""GPU idle-gap report from a rocprofv3 kernel trace:  python tools/gaps.py kernel_trace.csv [marker_kernel_substr] [top]

Splits the trace into steps at every launch of the marker kernel (default: fps_wave_k / fps_stream_k, one per FV2P step), and for
the steps after the first few prints: wall time, time with at least one kernel running (union over streams), idle time, and the
largest idle gaps with the kernels on either side — the places where the host (a .item(), a launch-bound run of small ops) keeps
the device waiting."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "fps_bbox_k"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2] and "prep" not in r[2] and "post" not in r[2]]
print(f"{len(rows)} kernels, {len(starts)} marker launches")
if len(starts) < 6:
    sys.exit(0)
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) // 2, len(starts) - 1)]
agg_gap = defaultdict(lambda: [0.0, 0])
tot_wall = tot_busy = 0.0
small = 0.0
for lo, hi in steps:
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], rows[hi][0]
    busy, end, last = 0.0, t0, seg[0][2]
    for s, e, name, _ in seg:
        if s > end:
            g = s - end
            key = (last[:70], name[:70])
            agg_gap[key][0] += g
            agg_gap[key][1] += 1
            busy += e - s
            end, last = e, name
        else:
            if e > end:
                busy += e - end
                end, last = e, name
    tot_wall += t1 - t0
    tot_busy += busy
n = len(steps)
print(f"steps analysed: {n}; wall {tot_wall / n / 1e6:.3f} ms/step, some kernel running {tot_busy / n / 1e6:.3f} ms/step, idle {(tot_wall - tot_busy) / n / 1e6:.3f} ms/step")
print("largest idle gaps (per step):   ms/step  count/step   after -> before")
for (a, b), (g, c) in sorted(agg_gap.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {g / n / 1e6:7.3f}  {c / n:6.1f}   {a}  ->  {b}")
hist = defaultdict(float)
for (a, b), (g, c) in agg_gap.items():
    avg = g / c
    bucket = "<5us" if avg < 5e3 else "<20us" if avg < 2e4 else "<100us" if avg < 1e5 else ">=100us"
    hist[bucket] += g / n / 1e6
print("idle by average gap size:", {k: round(v, 3) for k, v in hist.items()})
# per-stream busy time
per = defaultdict(float)
for lo, hi in steps:
    for s, e, name, q in rows[lo:hi]:
        per[q] += e - s
print("kernel time per stream/queue (ms/step):", {k: round(v / n / 1e6, 3) for k, v in per.items()})
# the busiest stream's own idle time: what it waits for (another stream's kernel running meanwhile, or nothing = the host)
main_q = max(per, key=per.get)
wait = defaultdict(lambda: [0.0, 0])
for lo, hi in steps:
    seg = rows[lo:hi]
    mine = [r for r in seg if r[3] == main_q]
    others = [r for r in seg if r[3] != main_q]
    for a, b in zip(mine, mine[1:]):
        g0, g1 = a[1], b[0]
        if g1 - g0 < 2000:
            continue
        cover, who = 0, "(no kernel on any stream: host)"
        for s_, e_, name, q in others:
            ov = min(e_, g1) - max(s_, g0)
            if ov > cover:
                cover, who = ov, name[:60]
        key = (who if cover > 0.5 * (g1 - g0) else "(no kernel on any stream: host)", b[2][:60])
        wait[key][0] += g1 - g0
        wait[key][1] += 1
tot = sum(v[0] for v in wait.values())
print(f"stream {main_q}: idle {tot / n / 1e6:.3f} ms/step in gaps >= 2 us;  ms/step  count/step  running elsewhere -> next kernel on this stream")
for (who, nxt), (g, c) in sorted(wait.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {g / n / 1e6:7.3f}  {c / n:6.1f}   {who}  ->  {nxt}")
