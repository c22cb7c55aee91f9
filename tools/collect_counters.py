#!/usr/bin/env python3
"""Reduce the counter passes of tools/prof_counters.sh to one JSON: per kernel the launches seen, the mean duration (kernel trace)
and the mean of every counter per launch.  Derived columns (MI355X_MICROARCH.md: SQ_* count quad-cycles summed over the SIMDs' waves,
GRBM_GUI_ACTIVE counts busy shader-clock cycles and rocprofv3 reports its SUM over the 8 XCDs -- a kernel that keeps every XCD busy
for its whole duration reads 8 x clock x duration; FETCH_SIZE x 2 for wide streaming reads, KB units):
    valu_per_wave            SQ_INSTS_VALU / SQ_WAVES
    clock_ghz_profiled       GRBM_GUI_ACTIVE / 8 / duration
    valu_issue_cycles_frac   4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   -- share of all SIMD cycles spent issuing VALU
    mean_waves_per_simd      4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
    hbm_bytes                FETCH_SIZE x 2048 + WRITE_SIZE x 1024
    python3 tools/collect_counters.py gpurun_out v1  ->  gpurun_out/counters_v1.json
    python3 tools/collect_counters.py --rederive profiles/r03/counters_v1.json     (derived columns again from the stored means)"""
import csv, glob, json, os, sys

N_XCD = 8


def derive(d):
    if d.get("SQ_WAVES"):
        d["valu_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
    if "GRBM_GUI_ACTIVE" in d and "duration_us_median" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / N_XCD
        d["clock_ghz_profiled"] = cyc / (d["duration_us_median"] * 1e3)
        if "SQ_ACTIVE_INST_VALU" in d:
            d["valu_issue_cycles_frac"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0)
        if "SQ_WAVE_CYCLES" in d:
            d["mean_waves_per_simd"] = 4.0 * d["SQ_WAVE_CYCLES"] / (cyc * 1024.0)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes"] = d["FETCH_SIZE"] * 2048.0 + d["WRITE_SIZE"] * 1024.0


if sys.argv[1] == "--rederive":
    res = json.load(open(sys.argv[2]))
    for d in res["kernels"].values():
        derive(d)
    res["note"] = "means per launch; separate rocprofv3 passes (kernel trace; two SQ passes; FETCH_SIZE; WRITE_SIZE); GRBM_GUI_ACTIVE is the sum over 8 XCDs"
    json.dump(res, open(sys.argv[2], "w"), indent=1)
    sys.exit(0)
root, tag = sys.argv[1], sys.argv[2]
KEEP = ("vit_lanes_kernel", "vit_octet_kernel", "vit_prep_ring4c_kernel", "vit_prep_ring4_kernel", "vit_prep_direct_kernel", "viterbi_kernel", "ofdm_demod_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            tail = name[name.index(k) + len(k):]
            return k + (tail[:tail.index(">") + 1] if tail.startswith("<") else "")
    return None


out = {}
for f in glob.glob(os.path.join(root, "cnt_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            d = out.setdefault(k, {"durations_us": []})
            d["durations_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for sub in ("cnt_sq", "cnt_sq2", "cnt_fetch", "cnt_write"):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {"durations_us": []})[c] = sum(v) / len(v)
            out[k]["launches_" + sub] = len(v)
for k, d in out.items():
    du = sorted(d.pop("durations_us"))
    if du:
        d["launches_trace"] = len(du)
        d["duration_us_median"] = du[len(du) // 2]
        d["duration_us_min"] = du[0]
    derive(d)
res = {"command": "tools/bench_decode.py --ensembles 4096 --steps 2 --no-overlap --spb 75", "note": __doc__.split("\n\n")[0] if False else
       "means per launch; separate rocprofv3 passes (kernel trace; two SQ passes; FETCH_SIZE; WRITE_SIZE); GRBM_GUI_ACTIVE is the sum over 8 XCDs", "kernels": out}
path = os.path.join(root, f"counters_{tag}.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res))
