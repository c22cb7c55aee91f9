#!/usr/bin/env python3
"""Several receivers in ONE process behind the drop-in C++ classes (tests/cpp/mirror_threads_driver in its timing mode): per receiver a reader
thread (OFDM_Demod::Process from memory, 65536-sample blocks), the demodulator's delivery thread and a radio thread that takes the frames from a
queue -- the reference app's OFDM / radio thread pair (examples/basic_radio_app.cpp:404-419) -- and calls FIC_Decoder + 18 MSC_Decoders.
Whole-process frames/s and real-time factor per receiver.

Every receiver reads its capture (--frames transmission frames) --loops times back to back: 2000 frames per receiver by default, so that the start of
a stream (first launches, first-use allocations, first touch of the page-locked buffers, the decoders' 16-CIF run-in: 0.1-0.2 s per process) is a few per cent of the run -- with 300 frames per receiver it was half of it and the figures were
about half the steady state's.  The wrap breaks nothing in these captures (they begin with the end of their last frame), so every loop delivers --frames frames.

    python tools/bench_mirror_multi.py [--receivers 1 2 4 8] [--frames 100] [--loops 20]
"""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def run_multi(torch, dabgpu, receivers=(1, 2, 4, 8), frames=100, loops=20, subchannels=18, modes=(None,), only_write=None):
    """modes: values of DABGPU_MIRROR_BANK to run every receiver count under (None = the classes' own rule, "0" = private pipelines, "1" = all members
    of the receiver bank); returns {"what", "runs": [...]} with a "bank" field per run (None when the driver binary has not been built)"""
    import dabsynth
    import resource
    driver = os.path.join(ROOT, "tests", "cpp", "mirror_threads_driver")
    if not os.path.exists(driver) and not only_write:
        return None
    dev = torch.device("cuda", 0)
    prs, mapper, _ = dabgpu.host_tables()
    out = {"what": __doc__.split("\n\n")[0], "runs": []}
    with tempfile.TemporaryDirectory() as d:
        if only_write:
            d = only_write
            os.makedirs(d, exist_ok=True)
        paths = []
        for k in range(max(receivers)):
            mux = dabsynth.Multiplex(1, 21 + k, dev)
            f2 = dabsynth.modulate(mux.frame_bits[0], prs, mapper)
            n = torch.arange(frames * dabsynth.NB_FRAME_SAMPLES + 2656 + 5000, device=dev, dtype=torch.float64)
            x = torch.cat([torch.zeros(5000 + 2656, dtype=torch.complex64, device=dev), f2.reshape(-1).repeat((frames + 1) // 2)[:frames * dabsynth.NB_FRAME_SAMPLES]])
            x = x * torch.polar(torch.ones_like(n), 2 * np.pi * (1.3e-3 - 4e-4 * k) * n).to(torch.complex64)
            x[:5000] = x[-5000:]
            x = x + 0.02 * torch.view_as_complex(torch.randn((x.numel(), 2), device=dev))
            p = os.path.join(d, f"rx{k}.c32")
            x.cpu().numpy().astype(np.complex64).tofile(p)
            paths.append(p)
            del x, n
        if only_write:
            return out
        args0 = [driver, "65536"]
        for s_ in range(subchannels):
            args0 += [str(48 * s_), "48", "2", "0"]

        def cgroup_stat():
            try:
                return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())}
            except (OSError, ValueError):
                return {}
        for mode in modes:
            env = dict(os.environ, DABGPU_DRIVER_BENCH="1", DABGPU_DRIVER_LOOPS=str(loops))
            if mode is not None:
                env["DABGPU_MIRROR_BANK"] = str(mode)
            if os.environ.get("DABGPU_DRIVER_PRELOAD"):                      # development: tools/exp/leakhist.c
                env["LD_PRELOAD"] = os.environ["DABGPU_DRIVER_PRELOAD"]
            # (DABGPU_LIBDIR: development -- another build of libdabgpu.so for a same-box A/B)
            env["LD_LIBRARY_PATH"] = os.environ.get("DABGPU_LIBDIR", os.path.join(ROOT, "dab-radio_amd")) + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
            for r in receivers:
                ru0, cg0 = resource.getrusage(resource.RUSAGE_CHILDREN), cgroup_stat()
                res = subprocess.run(args0 + ["--"] + paths[:r], capture_output=True, text=True, env=env, timeout=900)
                ru1, cg1 = resource.getrusage(resource.RUSAGE_CHILDREN), cgroup_stat()
                if res.returncode != 0:
                    raise RuntimeError("mirror_threads_driver failed: " + res.stderr[-2000:])
                run = json.loads(res.stdout.strip().splitlines()[-1])
                run["bank"] = env.get("DABGPU_MIRROR_BANK", "the classes' own rule")
                # what the process cost the host: CPU seconds of the whole child (start-up included) and how long the container's CPU quota throttled it
                cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
                run["host_cpu_seconds"] = round(cpu_s, 3)
                run["host_cpu_ms_per_frame"] = round(1e3 * cpu_s / max(1, run["frames"]), 3)
                if cg0 and cg1:
                    run["cgroup_throttled_ms"] = round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3, 1)
                    run["cgroup_periods_throttled"] = cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0)
                # DABGPU_MIRROR_PROFILE=1 / DABGPU_BANK_PROFILE=1 / DABGPU_DRIVER_CPU=1
                prof = [ln for ln in res.stderr.splitlines() if ln.startswith(("OFDM_Demod profile", "receiver bank", "LEAKHIST", "CPU by thread name"))]
                if prof:
                    run["profile"] = prof
                out["runs"].append(run)
    return out


if __name__ == "__main__":
    import torch, dabgpu
    ap = argparse.ArgumentParser()
    ap.add_argument("--receivers", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--loops", type=int, default=20)
    ap.add_argument("--subchannels", type=int, default=18)
    ap.add_argument("--only-write", default=None, help="write the IQ files rx<k>.c32 into this directory and stop (tools/timeline_multi.sh)")
    a = ap.parse_args()
    res = run_multi(torch, dabgpu, a.receivers, a.frames, a.loops, a.subchannels, modes=(os.environ.get("DABGPU_MIRROR_BANK"),), only_write=a.only_write)
    if res is None:
        print("tests/cpp/mirror_threads_driver has not been built (python -c 'import __graft_entry__ as g; g.build()')", file=sys.stderr); sys.exit(2)
    if not a.only_write:
        print(json.dumps(res))
