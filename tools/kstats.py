#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py file.csv n_steps [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / 1e6 / steps:.3f} ms over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step  calls/step {int(r['Calls']) / steps:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
