#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as 'avg us  calls  name' (names cut at the first '(')."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(f"{float(r['AverageNs']) / 1e3:10.1f} us x {int(r['Calls']):4d}  {r['Percentage']:>6}%  {r['Name'].split('(')[0][:90]}")
