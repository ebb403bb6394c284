#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel_stats.csv:  python tools/kstats.py FILE.csv STEPS [ROWS]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / steps / 1e3:.1f} us")
for r in rows[:top]:
    print(f"{r['Name'][:88]:88s} x{int(r['Calls']) / steps:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us  "
          f"per step {float(r['TotalDurationNs']) / steps / 1e3:7.1f}")
