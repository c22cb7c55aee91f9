#!/usr/bin/env python3
"""FIC Viterbi alone (dabgpu_fic_decode_frames) on F frames of random soft bits, per device mapping (DABGPU_VIT_MAP_*): the regime
between one wavefront per codeword and one lane per codeword.  Development bench; timing only (parity: tests/test_gpu_viterbi.py).
    python tools/bench_fic.py [--frames 4096] [--reps 20] [--mappings 1,2,3,0]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
import torch  # noqa: E402
import dabgpu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--mappings", default="1,2,3,0")
ap.add_argument("--no-check", action="store_true", help="ablation builds (DABGPU_LIB=build/exp/...) give wrong bytes on purpose")
args = ap.parse_args()
ctx = dabgpu.Context(0)
F = args.frames
g = torch.Generator(device="cuda"); g.manual_seed(1)
bits = torch.randint(-127, 128, (F, 230400), dtype=torch.int8, device="cuda", generator=g)
out = torch.zeros((F, 4, 96), dtype=torch.uint8, device="cuda")
res = torch.zeros((F * 4, 16), dtype=torch.uint8, device="cuda")
line = {"frames": F, "codewords": 4 * F, "trellis_steps": 4 * F * 774, "ms": {}, "sum": {}}
names = {0: "auto", 1: "wave", 2: "lane", 3: "octet"}
for m in [int(x) for x in args.mappings.split(",")]:
    ctx.viterbi_set_mapping(m)
    for _ in range(3):
        ctx.fic_decode_frames(bits, F, out, res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ctx.fic_decode_frames(bits, F, out, res)
    e1.record(); torch.cuda.synchronize()
    line["ms"][names[m]] = e0.elapsed_time(e1) / args.reps
    line["sum"][names[m]] = int(out.to(torch.int64).sum().item())
ctx.viterbi_set_mapping(0)
assert args.no_check or len(set(line["sum"].values())) == 1, "the mappings disagree"
line["lib"] = os.path.basename(dabgpu.LIB_PATH)
print(json.dumps(line))
