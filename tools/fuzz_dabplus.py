#!/usr/bin/env python3
"""Differential fuzz of the DAB+ outer-code kernel against the oracle: tests/test_gpu_dabplus.py's batch case (12 frame sizes, 0-6 damaged
symbols per codeword, uncorrectable codewords, damaged headers, a noise burst with re-acquisition, super frames straddling calls) with
many seeds -- every super-frame record and every corrected byte must equal the oracle's AAC_Frame_Processor restatement.
    python tools/fuzz_dabplus.py [--seeds 50] [--first 1000]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle as O  # noqa: E402
import dabgpu  # noqa: E402
import test_gpu_dabplus as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=50)
    ap.add_argument("--first", type=int, default=1000)
    a = ap.parse_args()
    O.build()
    ctx = dabgpu.Context(0)
    total = {"ok": 0, "rs_fail": 0, "fire_fail": 0, "wait": 0}
    for seed in range(a.first, a.first + a.seeds):
        seen = T.run_batch_of_streams(ctx, O, seed, strict_mix=False)      # asserts on the first difference
        for k in total:
            total[k] += seen[k]
    print(json.dumps({"seeds": a.seeds, "first_seed": a.first, "superframes_equal_oracle": total}))


if __name__ == "__main__":
    main()
