#!/usr/bin/env python3
"""Development bench: dabgpu_ofdm_demod_frames_history per soft-bit layout (natural / time-interleaver class order) into a frame-history
ring, F frames of random IQ.  DABGPU_LIB selects an A/B build (tools/build_exp.sh).
    python tools/bench_demod_layout.py [--frames 4096] [--reps 10]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
import torch, dabgpu
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--spb", type=int, default=0)
ap.add_argument("--data", default="randn", help="randn | ofdm (tools/dabsynth.py frames with carrier offsets)")
a = ap.parse_args()
ctx = dabgpu.Context(0)
F, H = a.frames, 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
if a.data == "ofdm":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dabsynth
    prs, mapper, _ = dabgpu.host_tables()
    iqc, _, freq = dabsynth.random_frames(F, 1000, torch.device("cuda", 0), mapper, prs)
    iq = torch.view_as_real(iqc).contiguous()
else:
    iq = torch.randn((F, 196608, 2), generator=g, dtype=torch.float32, device="cuda")
    freq = ((torch.rand(F, generator=g, device="cuda") * 2 - 1) * 2.0e-3).float()
hist = torch.zeros((F, H, 230400), dtype=torch.int8, device="cuda")
corr = torch.zeros((F, 76, 2), dtype=torch.float32, device="cuda")
fmt = dabgpu.IQ_FORMATS.index("raw_f32l")
out = {"frames": F, "lib": os.path.basename(dabgpu.LIB_PATH), "data": a.data, "spb": a.spb, "ms": {}}
for rnd in range(2):
    for name, layout in (("natural", 0), ("classed", 1)):
        for k in range(3):
            ctx.ofdm_demod_frames_history(iq, fmt, F, hist[:, k % H], freq_offset=freq, cp_corr=corr, symbols_per_block=a.spb, bits_frame_stride=H * 230400, bits_layout=layout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(a.reps):
            ctx.ofdm_demod_frames_history(iq, fmt, F, hist[:, k % H], freq_offset=freq, cp_corr=corr, symbols_per_block=a.spb, bits_frame_stride=H * 230400, bits_layout=layout)
        e1.record(); torch.cuda.synchronize()
        out["ms"][name] = e0.elapsed_time(e1) / a.reps
out["spb_chosen"] = ctx.ofdm_auto_symbols_per_block(F)
print(json.dumps(out))
