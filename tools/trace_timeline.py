#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap) of the last complete forward found in a rocprofv3
kernel_trace.csv:  python tools/trace_timeline.py gpurun_out/prof/.../NNN_kernel_trace.csv [n_kernels_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a forward starts at the fillBuffer (workspace memset); take the last-but-one complete group
starts = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-which - 1], starts[-which]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("mtmc::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}x{r['Grid_Size_Y']:>3s}  q{r['Queue_Id']} {name[:60]}")
    prev_end = max(prev_end, e)
    tot += e - s
print(f"forward span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us")
