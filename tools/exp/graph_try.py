import os, sys, json, time
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import dabgpu, bench
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = dabgpu.Context(0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    P = bench.Pipeline(ctx, dabgpu, torch, dev, E, min(E, 16), seed=7, inflight=1, synced=True)
torch.cuda.synchronize()
P.tune()
P.fill()
torch.cuda.synchronize()
def run_eager(n):
    t0 = time.perf_counter()
    for _ in range(n): P.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", run_eager(20), run_eager(50))
# capture one step per ring slot? the slot and frame number change every step: capture H*? steps = one full cycle of the ring and the mux period
H = P.H
nf = P.mux.n_frames
import math
cycle = H * nf // math.gcd(H, nf)
print("cycle", cycle, "H", H, "nf", nf)
st = P.streams[0]
g = torch.cuda.CUDAGraph()
# align j to a multiple of cycle
while P.j % cycle: P.step()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=st):
        for _ in range(cycle): P.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, 50 // cycle)
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", (time.perf_counter() - t0) / (reps * cycle) * 1e3)
    P.j += reps * cycle
    chk = P.check(dabgpu)
    print(json.dumps(chk)[:600])
except Exception as e:
    print("capture failed:", repr(e)[:500], dabgpu.last_error() if hasattr(dabgpu, "last_error") else "")
