#!/usr/bin/env python3
"""timeline of the dabgpu kernels in a rocprofv3 --kernel-trace csv (development tool): start offset, duration, queue, name
    python tools/ktimeline.py gpurun_out/prof/p_kernel_trace.csv [first] [count]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dabgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{a / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][8:60]}")
