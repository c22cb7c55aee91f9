#!/usr/bin/env python3
"""development: kernel timeline around the last pipelined steps of a `bench.py --workload full` trace (rocprofv3 --kernel-trace csv)
    python tools/exp/timeline_bg.py <kernel_trace.csv> [kernel-name substring that marks a step, default ofdm_demod] [rows]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dabgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "ofdm_demod"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
lanes = [i for i, r in enumerate(rows) if "vit_lanes" in r["Kernel_Name"]]
# the pipelined steps: lanes kernels whose time span overlaps a demodulation of the other queue; show around the middle of the run
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
i0 = max(0, idx[len(idx) // 2] - 10)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + n]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f us  +%8.1f  q%3s  %s" % (a / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][8:60]))
