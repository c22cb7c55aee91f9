#!/usr/bin/env python3
"""Development bench: dabgpu_ofdm_sync (coarse frequency + fine time synchronisation, 5 transforms per stream) on N streams.
    python tools/bench_sync.py [--streams 1024 4096]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
import torch
import dabgpu

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, nargs="+", default=[1024, 4096])
a = ap.parse_args()
ctx = dabgpu.Context(0)
out = {"lib": os.path.basename(dabgpu.LIB_PATH), "ms": {}}
for n in a.streams:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((n, 2552, 2), generator=g, dtype=torch.float32, device="cuda")
    st = torch.zeros((n, 8), dtype=torch.float32, device="cuda")
    for _ in range(5):
        ctx.ofdm_sync(x, n, 2552, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        ctx.ofdm_sync(x, n, 2552, st)
    e1.record(); torch.cuda.synchronize()
    out["ms"][n] = e0.elapsed_time(e1) / 50
print(json.dumps(out))
