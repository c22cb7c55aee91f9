#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 *_kernel_stats.csv: name (shortened), calls, average us, total ms
    python tools/kstats.py gpurun_out/prof/trace_kernel_stats.csv [n] [substring of the kernel name]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
if len(sys.argv) > 3:
    rows = [r for r in rows if sys.argv[3] in r["Name"]]
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} us {float(r['TotalDurationNs']) / 1e6:10.2f} ms")
