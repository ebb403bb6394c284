#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name:  python tools/pmc_summary.py <counter_collection.csv> [COUNTER]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0.0, 0])
for r in rows:
    if want and r["Counter_Name"] != want:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("mtmc::", "").replace("void ", "")
    key = (name, r["Counter_Name"], r.get("Grid_Size", ""))
    acc[key][0] += float(r["Counter_Value"])
    acc[key][1] += 1
print(f"{'kernel':52s} {'counter':12s} {'grid':>9s} {'calls':>6s} {'avg':>14s}")
for (name, ctr, grid), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{name[:52]:52s} {ctr:12s} {grid:>9s} {n:6d} {tot / n:14.1f}")
