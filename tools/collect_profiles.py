#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_round.sh (run on the GPU box) to the small summaries profiles/ keeps:
the kernel stats table, the first 40 rows of each counter dump, and hbm_traffic.json (FETCH_SIZE x 2 per
MI355X_MICROARCH.md's HBM section + WRITE_SIZE, KB units, averaged over the sampled launches of ofdm_demod_kernel).

    python3 tools/collect_profiles.py gpurun_out v5      # writes gpurun_out/summary_v5/
"""
import csv
import glob
import json
import os
import sys

ALGO_BYTES_PER_FRAME = 1803264
FRAMES = 1024


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    return hits[0] if hits else None


def counter_average(path, counter, kernel="ofdm_demod_kernel"):
    vals, head = [], []
    with open(path, newline="") as f:
        r = csv.reader(f)
        hdr = next(r)
        head.append(hdr)
        ik, ic, iv = hdr.index("Kernel_Name"), hdr.index("Counter_Name"), hdr.index("Counter_Value")
        for row in r:
            if kernel in row[ik] and row[ic] == counter:
                vals.append(float(row[iv]))
                if len(head) <= 40:
                    head.append(row)
    return (sum(vals) / len(vals) if vals else None), len(vals), head


def main():
    root, tag = sys.argv[1], sys.argv[2]
    rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
    out = os.path.join(root, f"summary_{tag}")
    os.makedirs(out, exist_ok=True)
    stats = find(os.path.join(root, "prof"), "*kernel_stats.csv")
    if stats:
        with open(stats) as f, open(os.path.join(out, f"kernel_stats_{tag}.csv"), "w") as g:
            g.write(f.read())
    traffic = {"frames_per_launch": FRAMES, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * FRAMES}
    for name, counter, key in (("pmc_fetch", "FETCH_SIZE", "fetch"), ("pmc_write", "WRITE_SIZE", "write")):
        path = find(os.path.join(root, name), "*counter_collection.csv")
        if not path:
            continue
        avg, n, head = counter_average(path, counter)
        with open(os.path.join(out, f"pmc_{key}_size_{tag}.csv"), "w", newline="") as g:
            csv.writer(g).writerows(head)
        traffic[f"{key}_kb_avg"] = avg
        traffic[f"{key}_launches_sampled"] = n
    if "fetch_kb_avg" in traffic and "write_kb_avg" in traffic and traffic["fetch_kb_avg"] and traffic["write_kb_avg"]:
        traffic["fetch_bytes_corrected_x2"] = traffic["fetch_kb_avg"] * 1024 * 2
        traffic["write_bytes"] = traffic["write_kb_avg"] * 1024
        traffic["bytes_per_launch"] = traffic["fetch_bytes_corrected_x2"] + traffic["write_bytes"]
        traffic["source"] = (f"profiles/{rnd}/pmc_fetch_size_{tag}.csv + pmc_write_size_{tag}.csv (separate rocprofv3 --pmc passes; "
                             "FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section; KB units)")
    with open(os.path.join(out, "hbm_traffic.json"), "w") as g:
        json.dump(traffic, g, indent=1)
    print(json.dumps(traffic))
    # channel decoder (tools/bench_decode.py --ensembles 1024, both history layouts): HBM bytes per launch of its kernels
    dec = {"ensembles": 1024, "note": "FETCH_SIZE x 2 + WRITE_SIZE (KB units) per launch, separate rocprofv3 --pmc passes of "
                                      "tools/bench_decode.py --ensembles 1024 --steps 2 --hist-layout <layout>"}
    for layout in ("classed", "natural"):
        per = {}
        for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            path = find(os.path.join(root, f"pmc_dec_{name}_{layout}"), "*counter_collection.csv")
            if not path:
                continue
            for kern in ("vit_prep_ring4c_kernel", "vit_prep_ring4_kernel", "vit_prep_direct_kernel", "vit_lanes_kernel<0, 1, 4>", "vit_lanes_kernel<0, 1, 5>",
                         "vit_lanes_kernel<0, 4, 4>", "vit_octet_kernel<0>", "ofdm_demod_kernel"):
                avg, n, _ = counter_average(path, counter, kern)
                if avg is not None:
                    per.setdefault(kern, {})[f"{name}_kb_avg"] = avg
                    per[kern][f"{name}_launches"] = n
        for kern, d in per.items():
            if "fetch_kb_avg" in d and "write_kb_avg" in d:
                d["hbm_bytes_per_launch"] = d["fetch_kb_avg"] * 2048 + d["write_kb_avg"] * 1024
        if per:
            dec[layout] = per
    if len(dec) > 2:
        with open(os.path.join(out, f"hbm_traffic_decode_{tag}.json"), "w") as g:
            json.dump(dec, g, indent=1)


if __name__ == "__main__":
    main()
