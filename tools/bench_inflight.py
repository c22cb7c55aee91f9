#!/usr/bin/env python3
"""Development bench: bench.py's Pipeline (configs[3]: demod -> FIC -> MSC per transmission frame of E ensembles) with 1 / 2 / 3 frames in
flight.   python tools/bench_inflight.py [--ensembles 4096]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")]
import torch
import dabgpu
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--ensembles", type=int, default=4096)
ap.add_argument("--reps", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {"ensembles": a.ensembles, "frames_per_s": {}}
for n in (1, 2, 3):
    ctx = dabgpu.Context(0)
    p = bench.Pipeline(ctx, dabgpu, torch, dev, a.ensembles, 64, seed=7, inflight=n, layout=1)
    p.fill()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.reps):
            p.step()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / a.reps
    chk = p.check(dabgpu)
    out["frames_per_s"][n] = a.ensembles / t
    out.setdefault("ok", {})[n] = bool(chk["fib_bytes_equal_transmitted"] and chk["msc_bytes_equal_transmitted"] and chk["fib_crc_pass"] == chk["fib_crc_expected"])
    del p, ctx
    torch.cuda.empty_cache()
print(json.dumps(out))
