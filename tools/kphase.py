#!/usr/bin/env python3
"""Phase clock of ofdm_demod_kernel (development tool): runs a -DDABGPU_EXP=8 build (tools/build_exp.sh ofdm_demod.hip e8=-DDABGPU_EXP=8)
and prints where a wavefront's core-clock cycles go per OFDM symbol, plus the core clock the kernel ran at.
    python tools/kphase.py [--lib build/exp/libdabgpu_e8.so] [--frames 1024] [--spb 25]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="build/exp/libdabgpu_e8.so")
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--spb", type=int, default=25)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
F = a.frames
dev = torch.device("cuda", 0)
iq = torch.randn((F, 196608, 2), dtype=torch.float32, device=dev)
freq = ((torch.rand(F, device=dev) * 2 - 1) * 2.4e-3).float()
bits = torch.empty((F, 230400), dtype=torch.int8, device=dev)
corr = torch.empty((F, 76, 2), dtype=torch.float32, device=dev)
chunks = (75 + a.spb - 1) // a.spb
dbg = torch.zeros((F * chunks * 4, 12), dtype=torch.float32, device=dev)
L = C.CDLL(os.path.join(ROOT, a.lib))
L.dabgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
L.dabgpu_ofdm_demod_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
ctx = C.c_void_p()
assert L.dabgpu_create(C.byref(ctx), 0, None, None) == 0
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(a.iters):
    if it == a.iters - 1: e0.record()
    assert L.dabgpu_ofdm_demod_frames(ctx, iq.data_ptr(), F, freq.data_ptr(), bits.data_ptr(), corr.data_ptr(), dbg.data_ptr(), None, a.spb, 0, stream) == 0
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.cpu().numpy().astype(np.float64)
names = ["PLL + corr + radix 4", "skew barrier", "exchange + row store + barrier", "prefetch issue + reads + radix 8", "transpose A + radix 8",
         "transpose B + radix 8", "demapper"]
nsym = d[:, 9]
per = d[:, :7] / nsym[:, None]
tot = d[:, 7] / nsym
clock = d[:, 7] / d[:, 8] * 100e6           # s_memrealtime ticks at 100 MHz
print(f"kernel {ms:.4f} ms (with stamps); core clock while running: median {np.median(clock)/1e9:.3f} GHz (p5 {np.percentile(clock,5)/1e9:.3f}, p95 {np.percentile(clock,95)/1e9:.3f})")
print(f"cycles per wave and symbol: median {np.median(tot):.0f}  mean {tot.mean():.0f}")
for k, n in enumerate(names):
    print(f"  {n:36s} mean {per[:, k].mean():7.0f}  median {np.median(per[:, k]):7.0f}  p95 {np.percentile(per[:, k], 95):7.0f}   {100 * per[:, k].mean() / tot.mean():5.1f} %")
life = d[:, 8] / 100e6 * 1e3
print(f"workgroup lifetime: median {np.median(life):.4f} ms, max {life.max():.4f} ms; start spread {(d[:,11].max()-d[:,11].min())/100e6*1e3:.4f} ms")
