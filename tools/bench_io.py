#!/usr/bin/env python3
"""Streaming rate of the data-format kernels (include/dabgpu.h) on one MI355X: achieved HBM GB/s
= (bytes read + bytes written) / kernel time, HIP events on torch's current stream.

    python tools/bench_io.py [--frames 512] [--reps 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))

import torch  # noqa: E402

import dabgpu  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    ctx = dabgpu.Context(0)
    n = args.frames * dabgpu.NB_FRAME_SAMPLES
    out = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    res = {"frames": args.frames, "iq_convert": {}, "bits": {}}
    for name in ("raw_u8", "raw_s16l", "raw_s16b", "wav_pcm24", "raw_s32l", "raw_f32b", "raw_f64l", "wav_alaw"):
        fmt = dabgpu.IQ_FORMATS.index(name)
        sb = dabgpu.iq_format_sample_bytes(fmt)
        raw = torch.randint(0, 256, (n * sb,), dtype=torch.uint8, device="cuda")
        ms = timed(lambda: ctx.iq_convert(raw, fmt, n, out), args.reps)
        res["iq_convert"][name] = {"ms": ms, "GB_per_s": n * (sb + 8) / ms / 1e6, "Msamples_per_s": n / ms / 1e3}
        del raw
    # demodulating straight from a capture format (fused loader) vs convert + demodulate
    bits = torch.empty((args.frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    corr = torch.empty((args.frames, 76, 2), dtype=torch.float32, device="cuda")
    res["demod"] = {}
    out.normal_()
    res["demod"]["c32"] = {"ms": timed(lambda: ctx.ofdm_demod_frames(out, bits, cp_corr=corr, n_frames=args.frames), args.reps)}
    for name in ("raw_u8", "raw_s16l", "raw_s16b"):
        fmt = dabgpu.IQ_FORMATS.index(name)
        sb = dabgpu.iq_format_sample_bytes(fmt)
        raw = torch.randint(0, 256, (n * sb,), dtype=torch.uint8, device="cuda")
        ms = timed(lambda: ctx.ofdm_demod_frames_raw(raw, fmt, args.frames, bits, cp_corr=corr), args.reps)
        res["demod"][name] = {"ms": ms, "frames_per_s": args.frames / ms * 1e3}
        del raw
    # transmission modes II-IV through the size-generic kernel (24 ms / 24 ms / 48 ms of signal per frame)
    res["modes"] = {}
    for mode in (2, 3, 4):
        p = dabgpu.ofdm_params(mode)
        F = 4 * args.frames
        x = torch.randn((F * p["nb_frame_samples"] * 2,), dtype=torch.float32, device="cuda")
        b = torch.empty((F, p["nb_frame_bits"]), dtype=torch.int8, device="cuda")
        ms = timed(lambda: ctx.ofdm_demod_frames_mode(mode, x, F, b), args.reps)
        algo = F * (p["nb_frame_samples"] * 8 + p["nb_frame_bits"])
        res["modes"][f"mode_{mode}"] = {"frames": F, "ms": ms, "frames_per_s": F / ms * 1e3, "GB_per_s": algo / ms / 1e6,
                                        "x_realtime": F / ms * 1e3 * p["nb_frame_samples"] / 2.048e6}
        del x, b
    nb = args.frames * dabgpu.NB_FRAME_BITS // 8
    soft = torch.randint(-128, 128, (8 * nb,), dtype=torch.int8, device="cuda")
    hard = torch.empty(nb, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: ctx.soft_bits_to_hard_bytes(soft, nb, hard), args.reps)
    res["bits"]["soft_to_hard"] = {"ms": ms, "GB_per_s": nb * 9 / ms / 1e6}
    ms = timed(lambda: ctx.hard_bytes_to_soft_bits(hard, nb, soft), args.reps)
    res["bits"]["hard_to_soft"] = {"ms": ms, "GB_per_s": nb * 9 / ms / 1e6}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
