#!/usr/bin/env python3
"""Development fuzz: dabgpu_stream_bank_process_retained against dabgpu_stream_bank_process_raw on the same streams with a RANDOM block
length per call (1,000 .. 600,000 samples: shorter than, around and longer than a frame), three capture formats, drop-outs and a
noise-only stream -- frame counts, every frame's soft bits and every status field after every call must be equal.  (The fixed-size
cases are tests/test_gpu_stream_bank.py::test_retained_blocks_*.)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import dabgpu
import oracle as O
from test_gpu_stream_bank import make_stream, noise_with_dips
ctx = dabgpu.Context(0)
for seed in range(6):
    rng = np.random.default_rng(seed)
    base = [make_stream(O, 100 + seed, 5, float(rng.uniform(-8e-3, 8e-3)), int(rng.integers(0, 3000)), float(rng.uniform(0.5, 6))),
            make_stream(O, 200 + seed, 5, float(rng.uniform(-8e-3, 8e-3)), int(rng.integers(0, 3000)), 2.0, dropout=(int(rng.integers(200000, 300000)), int(rng.integers(300000, 340000)))),
            noise_with_dips(300 + seed, streams_len=1100000)]
    n = min(s.size for s in base); E = len(base)
    name = ["raw_f32l", "raw_u8", "raw_s16l"][seed % 3]
    fmt = dabgpu.IQ_FORMATS.index(name)
    if name == "raw_f32l": q = np.stack([np.stack([s[:n].real, s[:n].imag], -1).astype(np.float32) for s in base])
    elif name == "raw_u8": q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], -1) / np.abs(s[:n]).max() * 127 + 127.5), 0, 255).astype(np.uint8) for s in base])
    else: q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], -1) / np.abs(s[:n]).max() * 30000), -32768, 32767).astype(np.int16) for s in base])
    cap = 600000; max_frames = cap // 191400 + 2
    banks = [dabgpu.StreamBank(ctx, E), dabgpu.StreamBank(ctx, E)]
    bufs = [torch.zeros((E, cap, 2), dtype=torch.from_numpy(q[:1, :1]).dtype, device="cuda") for _ in range(3)]
    bits = [torch.zeros((E, max_frames, 230400), dtype=torch.int8, device="cuda") for _ in range(2)]
    nf = [torch.zeros(E, dtype=torch.int32, device="cuda") for _ in range(2)]
    k, call, prev, frames = 0, 0, None, 0
    while k < n:
        m = int(min(n - k, rng.choice([rng.integers(1000, 70000), rng.integers(150000, 260000), rng.integers(260000, cap)])))
        cur = bufs[1 + call % 2]
        cur[:, :m] = torch.from_numpy(q[:, k:k + m]).cuda(); bufs[0][:, :m] = cur[:, :m]
        for b in bits: b.zero_()
        banks[0].process_raw(bufs[0], fmt, cap, m, bits[0], max_frames, nf[0])
        banks[1].process_retained(cur, fmt, cap, m, prev, bits[1], max_frames, nf[1])
        torch.cuda.synchronize()
        assert torch.equal(nf[0], nf[1]) and torch.equal(bits[0], bits[1]), (seed, k, m)
        s0, s1 = banks[0].status(), banks[1].status()
        for f in s0.dtype.names:
            a, b = (x[f].view(np.uint32) if x[f].dtype == np.float32 else x[f] for x in (s0, s1))
            assert np.array_equal(a, b), (seed, k, f)
        frames += int(nf[0].sum().item()); prev, call, k = cur, call + 1, k + m
    print("seed", seed, name, "calls", call, "frames", frames, "ok")
    for b in banks: b.close()
