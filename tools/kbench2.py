#!/usr/bin/env python3
"""Kernel A/B harness (development tool): times dabgpu_ofdm_demod_frames of SEVERAL builds of libdabgpu.so in one process,
interleaved rounds on the same device buffers, and compares output checksums against the first build.
    python tools/kbench2.py --libs build/exp/libdabgpu_base.so,build/exp/libdabgpu_e0.so [--data randn|ofdm|zeros]
Not a benchmark of record -- bench.py is."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--libs", type=str, required=True)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--spb", type=str, default="19")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--data", type=str, default="randn")
ap.add_argument("--pad", type=int, default=0, help="extra IQ samples between frames (only builds with -DDABGPU_EXP_PAD=<pad> read it)")
a = ap.parse_args()
F = a.frames
dev = torch.device("cuda", 0)
stride = 196608 + a.pad
if a.data == "ofdm":
    import dabgpu
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dabsynth
    prs, mapper, _ = dabgpu.host_tables()
    iqc, _, freq = dabsynth.random_frames(F, 1000, dev, mapper, prs)
    iq0 = torch.view_as_real(iqc).contiguous()
elif a.data == "zeros":
    iq0 = torch.zeros((F, 196608, 2), dtype=torch.float32, device=dev)
    freq = ((torch.rand(F, device=dev) * 2 - 1) * 2.4e-3).float()
else:
    iq0 = torch.randn((F, 196608, 2), dtype=torch.float32, device=dev)
    freq = ((torch.rand(F, device=dev) * 2 - 1) * 2.4e-3).float()
if a.pad:
    iq = torch.zeros((F, stride, 2), dtype=torch.float32, device=dev)
    iq[:, :196608] = iq0
    del iq0
else:
    iq = iq0
bits = torch.empty((F, 230400), dtype=torch.int8, device=dev)
corr = torch.empty((F, 76, 2), dtype=torch.float32, device=dev)
libs = []
for path in a.libs.split(","):
    L = C.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    L.dabgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
    L.dabgpu_ofdm_demod_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_size_t, C.c_void_p]
    ctx = C.c_void_p()
    assert L.dabgpu_create(C.byref(ctx), 0, None, None) == 0
    libs.append((os.path.basename(path), L, ctx))
stream = torch.cuda.current_stream().cuda_stream


def launch(L, ctx, s):
    rc = L.dabgpu_ofdm_demod_frames(ctx, iq.data_ptr(), F, freq.data_ptr(), bits.data_ptr(), corr.data_ptr(), None, None, s, 0, stream)
    assert rc == 0, rc


spbs = [int(s) for s in a.spb.split(",")]
res, ref = {}, None
for r in range(a.rounds):
    for name, L, ctx in libs:
        for s in spbs:
            launch(L, ctx, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                launch(L, ctx, s)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((name, s), []).append(e0.elapsed_time(e1) / a.iters)
            if r == 0:
                h = (int(bits.view(torch.int32).sum(dtype=torch.int64).item()), float(corr.double().sum().item()))
                if ref is None: ref = h
                elif h != ref: print(f"!! {name} spb {s}: output checksum differs {h} vs {ref}")
for (name, s), t in res.items():
    t = np.array(t)
    gbs = (196608 * 8 + 230400) * F / (np.median(t) * 1e-3) / 1e9
    print(f"{name:28s} spb {s:2d}: median {np.median(t):.4f} ms  min {t.min():.4f}  -> {gbs:7.1f} GB/s ({gbs/8000:.3f})   rounds " + " ".join(f"{x:.4f}" for x in t))
