"""In-situ launch times of the roofline kernel inside the FV2P step, from a rocprofv3 kernel trace of bench.py: per position of the launch
within a step (steps are delimited by the key-point sampler's launches) the average duration and what ran beside it on other streams.
    python tools/insitu_conv.py <kernel_trace.csv> [kernel substring]"""
import collections
import csv
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "conv_rows_ksplit<128, false, 64, 1"
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "fps_wave_k" in r[2]]
print(len(rows), "kernels,", len(marks), "steps")
acc = collections.defaultdict(list)
beside = collections.defaultdict(lambda: collections.Counter())
for a, b in zip(marks[len(marks) // 2:-1], marks[len(marks) // 2 + 1:]):
    seg = [(i, rows[i]) for i in range(a, b) if pat in rows[i][2]]
    for k, (i, r) in enumerate(seg):
        acc[(len(seg), k)].append((r[1] - r[0]) / 1e3)
        lo = i
        while lo > 0 and rows[lo - 1][1] > r[0] - 20_000_000:   # kernels that started up to 20 ms earlier may still run (the sampler)
            lo -= 1
        for j in range(lo, min(len(rows), i + 400)):
            if j == i:
                continue
            s, e, name = rows[j]
            if s >= r[1]:
                break
            ov = min(e, r[1]) - max(s, r[0])
            if ov > 0:
                beside[(len(seg), k)][name.split("(")[0][:70]] += ov / 1e3
for (n, k), v in sorted(acc.items()):
    top = ", ".join(f"{nm} {us / len(v):.0f} us" for nm, us in beside[(n, k)].most_common(3))
    print("launches/step %d  #%d  avg %.1f us  (n=%d)  beside it: %s" % (n, k, sum(v) / len(v), len(v), top))
