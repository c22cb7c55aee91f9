#!/usr/bin/env python3
"""bench.py -- Mode-I frames/s of the MI355X-native OFDM demodulation hot path.

Workload (BASELINE.json configs[1]): a batch of 1024 frame-aligned Mode-I frames of synthetic IQ per GPU,
resident in HBM as complex float32, pushed through the fused PLL + cyclic-prefix phase + 2048-pt FFT + DQPSK
+ frequency de-interleave + soft-bit kernel and the per-frame phase/fine-frequency tail.  One "step" = one
pass over the batch.  N>1: every rank owns an independent batch (independent ensembles, no collective in the
data path; torch.distributed is used only for the timing barrier and the max over ranks) -> weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--spb S] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (1,803,264 B/frame x frames)
/ mean launch duration of ofdm_demod_kernel measured with HIP events on the launch stream; `cpu_baseline` is
the oracle (a C port of the reference algorithm) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dab-radio_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dabgpu  # noqa: E402
from dabgpu import shard  # noqa: E402

ALGO_BYTES_PER_FRAME = 196608 * 8 + 230400          # c32 IQ read + int8 soft bits written
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8 TB/s spec
REALTIME_FRAMES_PER_S = 2.048e6 / 196608              # 10.4167


def synth_frames(n_frames, seed, device, mapper, prs, chunk=32):
    """Random-payload Mode-I frames built on the device with torch (plumbing, untimed): QPSK per ETSI 14.5 on
    de-interleaved positions, differential modulation from the PRS, IFFT, cyclic prefix, per-frame CFO.
    Returns (iq [n,196608] complex64 in frame-buffer layout, bits [n,75,3072] uint8, freq [n] float32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    mp = torch.from_numpy(mapper.astype(np.int64)).to(device)
    bins = torch.where(mp < 768, mp + (2048 - 768), mp - 768 + 1)            # carrier index -> FFT bin
    prs_t = torch.from_numpy(prs).to(device)
    a = 0.70710678
    iq = torch.zeros((n_frames, 196608), dtype=torch.complex64, device=device)
    bits = torch.empty((n_frames, 75, 3072), dtype=torch.uint8, device=device)
    # carrier frequency offset per frame (+-5 kHz at 2.048 MS/s) so the PLL does real work
    freq = ((torch.rand(n_frames, generator=g, device=device) * 2 - 1) * (5000.0 / 2.048e6)).float().contiguous()
    n = torch.arange(196608, device=device, dtype=torch.float64)
    for k0 in range(0, n_frames, chunk):
        k1 = min(k0 + chunk, n_frames)
        m = k1 - k0
        b = torch.randint(0, 2, (m, 75, 3072), generator=g, device=device, dtype=torch.uint8)
        bits[k0:k1] = b
        z = torch.complex((1.0 - 2.0 * b[:, :, :1536].float()) * a, (1.0 - 2.0 * b[:, :, 1536:].float()) * a)
        spec = torch.zeros((m, 76, 2048), dtype=torch.complex64, device=device)
        spec[:, 0] = prs_t
        cur = prs_t[bins].expand(m, -1).clone()
        for s_ in range(75):                                  # differential modulation, symbol by symbol
            cur = cur * z[:, s_]
            spec[:, s_ + 1, bins] = cur
        t = torch.fft.ifft(spec, dim=2) * 2048.0
        body = iq[k0:k1, : 76 * 2552].view(m, 76, 2552)
        body[:, :, 504:] = t
        body[:, :, :504] = t[:, :, 2048 - 504:]
        ph = (-2.0 * np.pi) * freq[k0:k1, None].double() * n[None, :]
        iq[k0:k1] *= torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
    iq *= (1.0 / 39.2)                                                          # unit-ish RMS like a normalised capture
    return iq, bits, freq


def cpu_baseline(seconds_target=12.0):
    """oracle (C port of the reference algorithm) on the host cores; bounded sample; returns dict for the JSON line"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import oracle as O
    O.lib()
    rng = np.random.default_rng(0)
    frames = []
    for _ in range(4):
        bits = rng.integers(0, 2, O.NB_FRAME_BITS, dtype=np.uint8)
        frames.append(O.tx_to_frame_buffer(O.apply_pll(O.modulate_frame(bits), 1.3e-4, 0.0)))
    frames = np.stack(frames)
    m = O.mapper()
    O.demod_frames_timing(frames, 2, -1.3e-4, m)                      # warm
    t0 = time.perf_counter()
    n1 = 64
    O.demod_frames_timing(frames, n1, -1.3e-4, m)
    dt1 = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))                          # cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))                                    # bounded sample: at most 64 worker threads

    def run(per_thread):
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(cores) as ex:                      # one GIL-free C call per thread
            list(ex.map(lambda i: O.demod_frames_timing(frames, per_thread, -1.3e-4, m), range(cores)))
        return time.perf_counter() - t0

    probe = run(8)                                                    # calibrate: oversubscribed hosts scale badly
    per_thread = int(max(8, min(8 * seconds_target / probe, 4 * seconds_target / (dt1 / n1))))
    dt = run(per_thread)
    done = per_thread * cores
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "host_logical_cpus": os.cpu_count(),
            "single_thread_value": n1 / dt1,
            "sample": f"{done} frame demods (PLL+CP-phase+76xFFT2048+DQPSK+demap) cycling 4 distinct synthetic frames, "
                      f"{cores} host threads x {per_thread} frames, oracle/dab_oracle_ofdm.c dab_demod_frame "
                      f"(FFTW absent -> oracle's own radix-4/8 FFT), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed steps before the W warm-up steps, until this much wall time has passed: MI355X raises its "
                         "clock over tens of ms of sustained load (0.59 -> 0.46 ms per launch, profiles/r01/ab_notes.md)")
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (BASELINE configs[1]: 1024)")
    ap.add_argument("--spb", type=int, default=0, help="data symbols per workgroup (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus N>1 must be launched through torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    ctx = dabgpu.Context(local_rank)
    prs, mapper, _ = dabgpu.host_tables()
    F = args.frames
    # weak scaling: the ensemble set grows with the number of GPUs; rank r owns the contiguous block shard_range gives
    first_unit, n_units = shard.shard_range(F * world, rank, world)
    assert n_units == F
    iq, tx_bits, freq = synth_frames(F, seed=1000 + first_unit, device=device, mapper=mapper, prs=prs)
    # the receiver corrects with the negative of the applied shift... the PLL multiplies by e^{+j2pi f n}
    d_freq = freq.clone()
    d_bits = torch.empty((F, 230400), dtype=torch.int8, device=device)
    d_corr = torch.empty((F, 76, 2), dtype=torch.float32, device=device)
    d_total = torch.empty(F, dtype=torch.float32, device=device)
    d_fine = torch.zeros(F, dtype=torch.float32, device=device)
    iq_f = torch.view_as_real(iq)

    def step():
        ctx.ofdm_demod_frames(iq_f, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=args.spb, n_frames=F)
        ctx.ofdm_phase_update(d_corr, F, total_phase=d_total, fine_freq=d_fine, beta=0.9)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:      # clock ramp-up, untimed and not part of W
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist if world > 1 else None, device)

    # ---- correctness of what was just timed (untimed) ----
    check = {}
    if not args.no_check:
        hard = (d_bits.view(F, 75, 3072) >= 0).to(torch.uint8)
        err = (hard != tx_bits)
        check["hard_bit_errors_vs_transmitted"] = int(err.sum().item())
        check["frames_with_hard_bit_errors"] = int((err.view(F, -1).sum(dim=1) > 0).sum().item())
        check["frames_checked"] = F
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            mism = 0
            for k in (0, F // 2, F - 1):
                exp = O.demod_frame(iq[k].cpu().numpy(), float(freq[k].item()))
                mism += int((exp["bits"] != d_bits[k].cpu().numpy()).sum())
            check["soft_bit_mismatches_vs_oracle_3_frames"] = mism

    # ---- dominant kernel alone: HIP events on the launch stream (torch's current stream) around n_ev back-to-back
    # launches, GPU already at its sustained clock (the timed region above has just run) ----
    n_ev = max(20, min(args.steps, 200))
    for _ in range(10):
        ctx.ofdm_demod_frames(iq_f, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=args.spb, n_frames=F)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n_ev):
        ctx.ofdm_demod_frames(iq_f, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=args.spb, n_frames=F)
    ev1.record()
    torch.cuda.synchronize()
    k_ms = float(ev0.elapsed_time(ev1)) / n_ev
    achieved = ALGO_BYTES_PER_FRAME * F / (k_ms * 1e-3) / 1e9

    if rank == 0:
        value = world * F * args.steps / elapsed
        line = {
            "metric": "dab_mode1_frames_per_sec", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batched 1024 Mode-I frames of synthetic IQ (c32, HBM-resident), "
                                   "PLL+CP-phase+FFT2048+DQPSK+demap, per GPU",
                       "frames_per_gpu_per_step": F, "symbols_per_block": args.spb or 19,
                       "sharding": "independent ensembles per rank, no data-path collective"},
            "x_realtime": value / REALTIME_FRAMES_PER_S,
            "roofline": {"bound": "hbm", "kernel": "ofdm_demod_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * F},
            "check": check,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        # PMC-derived HBM traffic per launch, when a profiles/ summary of this round exists (see profiles/README.md)
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
                tr = json.load(fh)
            if tr.get("frames_per_launch") == F:
                line["roofline"]["traffic"] = tr.get("bytes_per_launch")
                line["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
