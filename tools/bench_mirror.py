#!/usr/bin/env python3
"""Puts a number on the single-stream drop-in (the C++ mirror classes driven like basic_radio_app drives the reference's,
tests/cpp/mirror_harness in its timing mode): frames/s against the 10.42 frames/s of a live Mode-I signal, and the latency of every
FIC_Decoder::DecodeFIBGroup / MSC_Decoder::DecodeCIF call (one synchronous launch + two copies each).  One ensemble, the canonical
multiplex of 18 x 48 CU EEP 3-A sub-channels, unsynchronised stream with carrier offset and noise, read from a file in 65536-sample blocks
(the reference app's default block size).

    python tools/bench_mirror.py [--frames 40] [--subchannels 18]
"""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, dabgpu, dabsynth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--threads", type=int, default=9, help="decode threads of the second run")
ap.add_argument("--subchannels", type=int, default=18)
a = ap.parse_args()
dev = torch.device("cuda", 0)
prs, mapper, _ = dabgpu.host_tables()
mux = dabsynth.Multiplex(1, 21, dev)
frame = dabsynth.modulate(mux.frame_bits, prs, mapper)[0]                      # PRS + 75 symbols, NULL (zeros) last
n = torch.arange(a.frames * dabsynth.NB_FRAME_SAMPLES + 2656 + 5000, device=dev, dtype=torch.float64)
x = torch.cat([torch.zeros(5000 + 2656, dtype=torch.complex64, device=dev), frame.repeat(a.frames)])
x = x * torch.polar(torch.ones_like(n), 2 * np.pi * 1.3e-3 * n).to(torch.complex64)
x[:5000] = x[-5000:]                                                              # some signal before the first NULL
x = x + 0.02 * torch.view_as_complex(torch.randn((x.numel(), 2), device=dev))
harness = os.path.join(ROOT, "tests", "cpp", "mirror_harness")
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "iq.c32")
    x.cpu().numpy().astype(np.complex64).tofile(path)
    args = [harness, path, d, "65536"]
    for s in range(a.subchannels):
        args += [str(48 * s), "48", "2", "0"]
    runs = []
    for threads in (1, a.threads):
        env = dict(os.environ, DABGPU_HARNESS_BENCH="1", DABGPU_HARNESS_THREADS=str(threads))
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
        res = subprocess.run(args, capture_output=True, text=True, env=env, timeout=600)
        if res.returncode != 0:
            print(res.stderr[-2000:], file=sys.stderr)
            sys.exit(res.returncode)
        runs.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]))
out = runs[0]
out["what"] = "OFDM_Demod::Process + 4 x DecodeFIBGroup + 4 x %d x DecodeCIF per frame, all synchronous, one caller thread" % a.subchannels
out["with_decode_threads"] = dict(runs[1], what="the same with the sub-channels of a CIF decoded by %d threads, one task per sub-channel as "
                                  "basic_radio's thread pool runs them (each MSC_Decoder owns a device context)" % a.threads)
print(json.dumps(out))
