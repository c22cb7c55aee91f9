#!/usr/bin/env python3
"""Kernel A/B harness (development tool): times dabgpu_ofdm_demod_frames for the kernel variants selected by
the DABGPU_VARIANT environment switch, interleaved rounds in ONE process, random-noise IQ of realistic power.
Not a benchmark of record -- bench.py is."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
import numpy as np, torch, dabgpu, ctypes as C

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--variants", type=str, default="0")
ap.add_argument("--spb", type=str, default="19")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--data", type=str, default="randn", help="randn | ofdm (bench.py synthetic frames) | zeros")
a = ap.parse_args()
ctx = dabgpu.Context(0)
F = a.frames
if a.data == "ofdm":
    sys.path.insert(0, ROOT)
    src = open(os.path.join(ROOT, "bench.py")).read().split("def cpu_baseline")[0]
    ns = {"__file__": os.path.join(ROOT, "bench.py")}
    exec(compile(src, "bench_head", "exec"), ns)
    prs, mapper, _ = dabgpu.host_tables()
    iqc, _, freq = ns["synth_frames"](F, 1000, torch.device("cuda", 0), mapper, prs)
    iq = torch.view_as_real(iqc).contiguous()
elif a.data == "zeros":
    iq = torch.zeros((F, 196608, 2), dtype=torch.float32, device="cuda")
    freq = ((torch.rand(F, device="cuda") * 2 - 1) * 2.4e-3).float()
else:
    iq = torch.randn((F, 196608, 2), dtype=torch.float32, device="cuda")
    freq = ((torch.rand(F, device="cuda") * 2 - 1) * 2.4e-3).float()
bits = torch.empty((F, 230400), dtype=torch.int8, device="cuda")
corr = torch.empty((F, 76, 2), dtype=torch.float32, device="cuda")
L = dabgpu.lib()
L.dabgpu_debug_set_variant.argtypes = [C.c_int]
variants = [int(v) for v in a.variants.split(",")]
spbs = [int(s) for s in a.spb.split(",")]
ref = None
res = {}
for r in range(a.rounds):
    for v in variants:
        for s in spbs:
            L.dabgpu_debug_set_variant(v)
            ctx.ofdm_demod_frames(iq, bits, freq_offset=freq, cp_corr=corr, symbols_per_block=s, n_frames=F)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ctx.ofdm_demod_frames(iq, bits, freq_offset=freq, cp_corr=corr, symbols_per_block=s, n_frames=F)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((v, s), []).append(e0.elapsed_time(e1) / a.iters)
            if r == 0:
                h = (int(bits.view(torch.int32).sum(dtype=torch.int64).item()), float(corr.double().sum().item()))
                if ref is None: ref = h
                elif h != ref: print(f"!! variant {v} spb {s} output checksum differs {h} vs {ref}")
for (v, s), t in sorted(res.items()):
    t = np.array(t)
    gbs = (196608 * 8 + 230400) * F / (np.median(t) * 1e-3) / 1e9
    print("   rounds:", " ".join(f"{x:.4f}" for x in t))
    print(f"variant {v:2d} spb {s:2d}: median {np.median(t):.4f} ms  min {t.min():.4f}  -> {gbs:7.1f} GB/s  {F/np.median(t)*1e3/1e6:.3f} Mframes/s")
