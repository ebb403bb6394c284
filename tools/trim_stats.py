#!/usr/bin/env python3
"""Keep a rocprofv3 kernel_stats.csv readable: kernel names cut at the first '(' / 100 characters, torch's helper kernels
folded into one 'other (torch / rocprim / runtime)' row.   python tools/trim_stats.py IN.csv OUT.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keep, other = [], {"Calls": 0, "TotalDurationNs": 0.0}
for r in rows:
    name = r["Name"]
    if "mtmc::" in name or "_ZN4mtmc" in name:
        r["Name"] = ("mtmc::split_rows_kernel" if "split_rows_kernel" in name else name.split("(")[0].replace("void ", ""))[:100]
        keep.append(r)
    else:
        other["Calls"] += int(r["Calls"])
        other["TotalDurationNs"] += float(r["TotalDurationNs"])
tot = sum(float(r["TotalDurationNs"]) for r in keep) + other["TotalDurationNs"]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in keep:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                    f"{100 * float(r['TotalDurationNs']) / tot:.2f}", r["MinNs"], r["MaxNs"]])
    if other["Calls"]:
        w.writerow(["other (torch / rocprim / runtime kernels: data generation, memset, copies)", other["Calls"],
                    other["TotalDurationNs"], other["TotalDurationNs"] / other["Calls"],
                    f"{100 * other['TotalDurationNs'] / tot:.2f}", "", ""])
