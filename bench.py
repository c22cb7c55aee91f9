#!/usr/bin/env python3
"""bench.py -- Mode-I frames/s of the MI355X-native DAB receive path (driver contract: one JSON line from rank 0).

Workloads
  demod (default; BASELINE.json configs[1], the configuration the metric is quoted on): a batch of 1024 frame-aligned Mode-I
        frames of synthetic IQ per GPU, resident in HBM as complex float32, through the fused PLL + cyclic-prefix phase +
        2048-pt FFT + DQPSK + frequency de-interleave + soft-bit kernel and the per-frame phase / fine-frequency tail.
        One "step" = one pass over the batch.  At N = 1 the line also carries `extra.configs2` / `extra.configs3`
        (demod + FIC Viterbi, and full FIC + MSC for 4096 concurrent ensembles) with their own roofline blocks -- the path SURVEY 8(d)
        defines for them: every receiver has its own carrier offset (+-5 kHz) and timing offset (+-100 samples), and every frame runs
        PRS synchronisation (coarse + fine) -> demodulation where and with the offset the synchroniser found -> fine-frequency update
        (dabgpu_ofdm_sync_demod_frames) -> FIC / MSC decode -- and `extra.chain`: 4096 UNSYNCHRONISED raw_u8 streams through the
        device-resident chain stream bank -> history rings -> FIC + MSC -> DAB+ outer code (tools/bench_chain.py).
  full  (BASELINE.json configs[4], per GPU): 8192 ensembles per GPU (built on the device from <= 64 seeded multiplexes),
        one step = demodulate one transmission frame of every ensemble into its frame-history ring + FIC Viterbi (4 FIB groups)
        + MSC time de-interleave, Viterbi and descrambling of 18 sub-channels x 4 CIFs.

N > 1: every rank owns an independent block of frames / ensembles (dabgpu.shard.shard_range), no collective in the data path;
torch.distributed (RCCL) carries only the timing barrier and the max over ranks -> "scaling": "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload demod|full]
`python bench.py --gpus N` started plainly (no RANK in the environment) launches N ranks itself through
torch.distributed.run -- as a child process, before this process touches the GPU -- and exits with the child's status.

`roofline` (dominant kernel ofdm_demod_kernel): achieved = algorithmic bytes per launch (1,803,264 B/frame x frames) /
mean launch duration measured with ONE pair of HIP events around the back-to-back launches of the timed loop, on their stream;
`cpu_baseline` = the oracle (C port of the reference algorithm) timed on this box's host cores on a bounded sample of the metric's
workload; `cpu_baseline_full` = the oracle's whole receive chain (sync + demod + FIC + 18 x MSC per frame) beside configs[2]/[3].
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

ALGO_BYTES_PER_FRAME = 196608 * 8 + 230400          # c32 IQ read + int8 soft bits written
NULL_SYMBOL_BYTES = 2656 * 8                          # the null symbol is part of SURVEY 8(d)'s figure but no demodulator reads it
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: 8 TB/s spec
REALTIME_FRAMES_PER_S = 2.048e6 / 196608              # 10.4167
# Viterbi kernels: VALU-issue bound.  peak trellis steps/s = SIMDs x clock x codewords per wavefront /
# (VALU instructions per wavefront-step x cycles per instruction); instruction counts from the ISA of this build
# (tools/isa_count.py), 4 cycles per packed-integer / cross-lane instruction (tools/ubench/pk16_rate.hip), 2.4 GHz, 1024 SIMDs.
# Round 3: the two batch mappings are priced with the MEASURED instruction count of a whole wavefront (prologue, forward pass,
# chain-back, CRC) -- SQ_INSTS_VALU / SQ_WAVES / trellis steps per codeword of profiles/r03/counters_v1.json:
#   vit_lanes_kernel  280790.7 / 1542 = 182.1 (its forward loop alone is 169 in the ISA), vit_octet_kernel 31036.1 / 774 = 40.1
VIT_LANES_INSTR_PER_STEP = 182.1
VIT_OCTET_INSTR_PER_STEP = 40.1
VIT_WAVE_INSTR_PER_STEP = 23.0
VIT_CYCLES_PER_INSTR = 4.0
N_SIMD, CLOCK_HZ = 1024, 2.4e9
# instruction-count-free bound of the trellis recursion (whatever the mapping): 32 butterflies x 8 packed-u16 operations (4 adds, 2 mins,
# 2 saturated differences for the decisions) / 2 states per packed operation = 128 issue-4 instructions per step of 64 codewords
VIT_ALGO_INSTR_PER_64_STEPS = 128.0
VIT_ALGO_PEAK_GSTEPS = N_SIMD * CLOCK_HZ * 64.0 / (VIT_ALGO_INSTR_PER_64_STEPS * VIT_CYCLES_PER_INSTR) / 1e9      # 307.2
# sync-enabled configurations: a receiver's slice of a frame = LEAD samples of the preceding NULL symbol, then the frame; the PRS is
# expected at LEAD and found within +-100 samples of it
SLICE_LEAD = 1024
SLICE_SAMPLES = SLICE_LEAD + 1544 + 196608


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("demod", "full"), default="demod")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed steps before the W warm-up steps, until this much wall time has passed: MI355X settles its "
                         "clock over tens of ms of sustained load (profiles/r01/ab_notes.md)")
    ap.add_argument("--frames", type=int, default=1024, help="demod: frames per GPU per step (BASELINE configs[1]: 1024)")
    ap.add_argument("--ensembles", type=int, default=8192, help="full: ensembles per GPU (BASELINE configs[4]: 65536 / 8)")
    ap.add_argument("--distinct", type=int, default=64, help="full / extras: distinct seeded multiplexes the ensembles are built from")
    ap.add_argument("--extra-ensembles", type=int, default=4096, help="demod, N = 1: ensembles of extra.configs2 / configs3")
    ap.add_argument("--inflight", type=int, default=2, help="full: transmission frames in flight (one stream + context each)")
    ap.add_argument("--hist-layout", choices=("classed", "natural"), default="classed",
                    help="full / extras: order of the MSC soft bits in the frame-history ring (classed = DABGPU_BITS_MSC_CLASSED)")
    ap.add_argument("--spb", type=int, default=0, help="data symbols per workgroup (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-mirror", action="store_true", help="demod, N = 1: skip extra.one_receiver (tests/cpp/mirror_harness: one receiver behind the C++ classes)")
    ap.add_argument("--no-mixed", action="store_true", help="demod, N = 1: skip extra.configs3_mixed (configs[3] on a heterogeneous multiplex at --extra-ensembles)")
    ap.add_argument("--no-chain", action="store_true", help="demod, N = 1: skip extra.chain (the unsynchronised-stream chain at --extra-ensembles)")
    ap.add_argument("--lanes", choices=("alternate", "split"), default="alternate",
                    help="full / extras with two frames in flight: frame j wholly on stream j mod 2, or all demodulations on one (high-priority) stream and all decodes on the other")
    ap.add_argument("--aligned", action="store_true",
                    help="full / extras: frame-aligned input with synchronisation bypassed (what rounds 1-3 timed) instead of the sync-enabled path")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / sharding plumbing only, on CPU with gloo (tests/test_bench_launcher.py); no kernels, value = null")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="with --dry-run: this rank exits 3 (exit-status propagation test)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 400 if args.workload == "demod" else 10
    if args.warmup is None:
        args.warmup = 50 if args.workload == "demod" else 2
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a rendezvous in the environment: start the N ranks as a child (never re-exec a process
    that may have touched the GPU) and hand its exit status back."""
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def host_cores():
    """One logical CPU per PHYSICAL core of this process's affinity mask (both sockets), from the kernel's topology files: the workers of the CPU
    baselines are pinned to these.  Returns (cpus [int], dict for the JSON line)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen, cpus, packages = set(), [], set()
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                sib = fh.read().strip()
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as fh:
                pkg = fh.read().strip()
        except OSError:
            sib, pkg = str(c), "0"
        key = (pkg, sib)
        if key in seen:
            continue
        seen.add(key)
        packages.add(pkg)
        cpus.append(c)
    topo = {"physical_cores_in_affinity_mask": len(cpus), "logical_cpus_in_affinity_mask": len(allowed), "sockets": len(packages),
            "host_logical_cpus": os.cpu_count(), "cgroup_cpu_quota": None}
    # a container may see every CPU of the host and still be allowed only N CPUs' worth of time (cgroup v2 cpu.max / v1 cfs quota): more runnable
    # threads than that are throttled, they do not run.  The workers are then N physical cores, not all of them (the GPU boxes of this pool: 256
    # logical CPUs visible, quota 16)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = float(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = float(fh.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        topo["cgroup_cpu_quota"] = quota
        n = max(1, int(quota))
        if n < len(cpus):
            step = len(cpus) / n                                      # spread over the mask (both sockets), one worker per chosen core
            cpus = [cpus[int(k * step)] for k in range(n)]
    topo["workers"] = len(cpus)
    return cpus, topo


def run_pinned(cpus, work):
    """work(i) on one thread per entry of cpus, each pinned to its CPU (sched_setaffinity(0) binds the CALLING thread on Linux), all released
    together; the work is ONE GIL-free C call per thread.  Returns (results, seconds from release to the last thread's end)."""
    import threading
    n = len(cpus)
    res, errs = [None] * n, []
    ready, go = threading.Barrier(n + 1), threading.Event()

    def body(i):
        try:
            try:
                os.sched_setaffinity(0, {cpus[i]})
            except (AttributeError, OSError):
                pass
            ready.wait()
            go.wait()
            res[i] = work(i)
        except Exception as ex:                                      # noqa: BLE001 -- reported by the caller
            errs.append(ex)
    ths = [threading.Thread(target=body, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    ready.wait()
    t0 = time.perf_counter()
    go.set()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    return res, dt


def cpu_baseline(seconds_target=12.0):
    """oracle (C port of the reference algorithm) on the host cores; bounded sample; returns dict for the JSON line"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import numpy as np
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
    cpus, topo = host_cores()                                         # one worker per physical core of the affinity mask, pinned (both sockets)
    cores = len(cpus)

    def run(per_thread):
        _, dt_ = run_pinned(cpus, lambda i: O.demod_frames_timing(frames, per_thread, -1.3e-4, m))
        return dt_

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
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "threads": cores, "kind": "port", "cpu_model": cpu_model,
            "topology": topo, "calibration": cpu_calibration(),
            "single_thread_value": n1 / dt1, "scaling_efficiency": (done / dt) / (cores * (n1 / dt1)),
            "scaling_note": "value / (cores x single-thread value); cores = the physical cores of the affinity mask, or the container's CPU quota when that is "
                            "smaller (topology.cgroup_cpu_quota: more runnable threads than the quota are throttled); the single thread runs alone at its boost "
                            "clock, the workers together at the all-core clock",
            "sample": f"{done} frame demods (PLL+CP-phase+76xFFT2048+DQPSK+demap) cycling 4 distinct synthetic frames, "
                      f"{cores} pinned worker threads (one per physical core) x {per_thread} frames, oracle/dab_oracle_ofdm.c dab_demod_frame "
                      f"(FFTW absent -> oracle's own radix-4/8 FFT), {dt:.1f} s wall"}


def cpu_calibration():
    """port / reference time ratios on the parts of the reference that compile (tools/cpu_calibration.py, measured in the build container where
    /root/reference exists; the committed summary travels with the repository): how to read a `kind: "port"` baseline"""
    try:
        with open(os.path.join(ROOT, "profiles", "r05", "cpu_calibration.json")) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None
    return {"source": "profiles/r05/cpu_calibration.json (tools/cpu_calibration.py: reference objects compiled in place vs this port, one thread, build container)",
            "port_time_over_reference_time": {r["piece"]: r["port_time_over_reference_time"] for r in doc.get("rows", [])},
            "not_calibrated": doc.get("not_calibrated")}


def hbm_roofline(kernel, k_ms, frames):
    achieved = ALGO_BYTES_PER_FRAME * frames / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "achieved_excl_null": achieved * (1.0 - NULL_SYMBOL_BYTES / ALGO_BYTES_PER_FRAME),
            "traffic": None, "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * frames}


def _counter_evidence():
    """VALU instructions per wavefront and the clock under the profiler of the batch Viterbi kernels, from the newest
    profiles/r*/counters_v*.json (tools/prof_counters.sh: rocprofv3 --pmc passes at 4096 ensembles); {} when no summary is there.
    Keys: codewords per wavefront (64 = lane mapping, 8 = octet mapping); any instantiation of the kernel found in the summary counts
    (the one with the most wavefronts = the one the 4096-ensemble run was dominated by)."""
    import glob
    import re
    out = {}
    try:
        def order(p):
            m = re.search(r"profiles/r(\d+)/counters_v(\d+)\.json$", p.replace(os.sep, "/"))
            return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "counters_v*.json")), key=order)[-1]
        with open(path) as fh:
            k = json.load(fh)["kernels"]
        for prefix, steps, key in (("vit_lanes_kernel", 1542.0, 64.0), ("vit_octet_kernel", 774.0, 8.0)):
            cands = [(n, v) for n, v in k.items() if n.startswith(prefix) and v.get("valu_per_wave")]
            if not cands:
                continue
            name, v = max(cands, key=lambda nv: nv[1].get("SQ_WAVES", 0) or 0)
            out[key] = {"instr_per_step": v["valu_per_wave"] / steps, "clock_ghz_profiled": v.get("clock_ghz_profiled"),
                        "valu_issue_cycles_frac": v.get("valu_issue_cycles_frac"), "source": os.path.relpath(path, ROOT), "kernel": name}
    except Exception:
        pass
    return out


def viterbi_roofline(kernel, steps, k_ms, lanes):
    """Two bounds of the trellis recursion, both VALU issue (no dense contraction, far from HBM):
      frac               against the ALGORITHM's bound: 128 issue-4 packed-u16 instructions per step of 64 codewords -> 307 G trellis steps/s,
                         whatever the mapping and however many instructions the kernel at hand spends (decision gather, de-puncturing,
                         chain-back, CRC all count against it)
      issue_efficiency   against the kernel's OWN measured instruction stream (SQ_INSTS_VALU / SQ_WAVES / steps of the newest counter
                         summary under profiles/, else the ISA count): how close the wavefronts come to back-to-back issue
    `lanes`: codewords per wavefront of the mapping -- 64 (one lane per codeword), 8 (eight lanes per codeword) or 1 / False (one wavefront
    per codeword)."""
    per_wave = float(lanes) if lanes else 1.0
    if lanes is True:
        per_wave = 64.0
    instr = {64.0: VIT_LANES_INSTR_PER_STEP, 8.0: VIT_OCTET_INSTR_PER_STEP}.get(per_wave, VIT_WAVE_INSTR_PER_STEP)
    ev = _counter_evidence().get(per_wave)
    if ev:                                            # measured: SQ_INSTS_VALU / SQ_WAVES / trellis steps per codeword
        instr = round(ev["instr_per_step"], 2)
    own_peak = N_SIMD * CLOCK_HZ * per_wave / (instr * VIT_CYCLES_PER_INSTR) / 1e9
    achieved = steps / (k_ms * 1e-3) / 1e9
    out = {"bound": "valu_issue", "kernel": kernel, "achieved": achieved, "peak": VIT_ALGO_PEAK_GSTEPS, "unit": "G trellis steps/s",
           "frac": achieved / VIT_ALGO_PEAK_GSTEPS, "kernel_ms": k_ms, "trellis_steps_per_launch": steps,
           "peak_definition": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9} GHz x 64 codewords / ({int(VIT_ALGO_INSTR_PER_64_STEPS)} packed-u16 add / min / "
                              f"saturated-difference instructions per step of 64 codewords x {VIT_CYCLES_PER_INSTR} cycles): the algorithm's bound",
           "issue_efficiency": {"frac": achieved / own_peak, "peak": own_peak,
                                "peak_definition": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9} GHz x {int(per_wave)} codewords per wavefront / ({instr} VALU "
                                                   f"instructions per wavefront-step of THIS kernel x {VIT_CYCLES_PER_INSTR} cycles)"}}
    if per_wave == 64.0:
        # where the instructions of a trellis step go, and what each part costs in launch time (timing-only variants, tools/abl_lanes.py)
        try:
            with open(os.path.join(ROOT, "profiles", "r05", "abl_vit_lanes.json")) as fh:
                ab = json.load(fh)
            out["instr_per_step"] = ab["instr_per_step"]
            out["ablation"] = {"source": "profiles/r05/abl_vit_lanes.json (tools/abl_lanes.py, 4096 ensembles, gather + trellis call)",
                               "ms": {t: round(v["msc_call_ms"], 3) for t, v in ab["variants"].items() if not t.startswith("stagger")},
                               "reading": ab["reading"]}
        except (OSError, ValueError, KeyError):
            pass
    if ev:
        out["issue_efficiency"]["counters"] = {"source": ev["source"], "kernel": ev["kernel"], "valu_instructions_per_wavefront_step": ev["instr_per_step"],
                                               "clock_ghz_under_profiler": ev["clock_ghz_profiled"],
                                               "valu_issue_share_of_simd_cycles": ev["valu_issue_cycles_frac"]}
    return out


class Pipeline:
    """E ensembles, one transmission frame of IQ each, frame-history ring of H slots, FIC + MSC outputs (configs[2]/[3]/[4]).
    A step decodes the FIC and the MSC of a frame with ONE call (dabgpu_decode_frames_layout): the 4 FIB groups of every ensemble join the
    MSC's trellis launch, whose last round of wavefront slots the MSC's own groups do not fill.

    synced = True (the path SURVEY 8(d) defines for configs 3 / 4 / 5): receiver e has its own carrier offset (uniform in +-5 kHz) and its
    frames begin toff[e] (uniform in +-100) samples away from where it expects them; every frame goes through
    dabgpu_ofdm_sync_demod_frames -- RunCoarseFreqSync + RunFineTimeSync on the expected PRS position, demodulation from the position and
    with the offset (coarse + fine) just tracked, fine-frequency update from the cyclic-prefix phase -- with the per-receiver sync records
    resident on the device from frame to frame (ofdm_demodulator.cpp:360-548, :650-766, :606-618).  synced = False: frame-aligned input,
    no carrier offset, synchronisation bypassed (what rounds 1-3 timed; an upper bound).

    `inflight` frames are in flight at once, frame j on stream / context j mod inflight (a context owns its scratch, so concurrent
    calls need one each): the next frame's HBM-bound demodulation and gather kernels fill the wavefront slots that the trellis kernel's
    last, partial round leaves idle.  Dependencies kept with events: msc(j) reads the ring slots of frames j-4..j -> waits for
    demod(j-1), ...; demod(j) overwrites the slot of frame j-H, last read by msc(j-H+4) -> waits for it; synced: the synchroniser of
    frame j reads the fine-frequency word frame j-1's phase tail wrote -> demod(j) waits for demod(j-1)."""

    def __init__(self, ctx, dabgpu, torch, device, E, n_distinct, seed, inflight=1, layout=1, synced=True, noise=0.05, lanes="alternate", mux_layout=None):
        import dabsynth
        self.torch, self.E, self.inflight, self.synced, self.dabgpu = torch, E, inflight, synced, dabgpu
        # lanes = "alternate": frame j entirely on stream j mod inflight; "split" (inflight = 2): every demodulation on stream 0 (high
        # priority), every decode on stream 1 -- the front end of frame j + 1 is enqueued behind that of frame j, not behind j's decode
        self.lanes = lanes if inflight == 2 else "alternate"
        # layout of the MSC soft bits in the history ring: 1 = time-interleaver class order (DABGPU_BITS_MSC_CLASSED: the demodulator
        # writes it for free and the decoder's gather then reads ~1.3 instead of 4.75 history bytes per soft bit), 0 = On_OFDM_Frame()
        self.layout, self.fmt_f32 = layout, dabgpu.IQ_FORMATS.index("raw_f32l")
        self.H = 5 if inflight == 1 else 8
        prs, mapper, _ = dabgpu.host_tables()
        # two stored transmission frames that repeat (8 CIFs of changing payload, time interleaved): decoded bytes then prove WHICH
        # ring slots / ages / frames in flight they came from (tools/dabsynth.py)
        # mux_layout: the multiplex every ensemble carries (tools/dabsynth.py: canonical_layout() = 18 x 48 CU EEP 3-A by default, mixed_layout())
        self.iq, self.mux = dabsynth.ensemble_iq(E, min(n_distinct, E), seed, device, mapper, prs, noise=0.0 if synced else noise, layout=mux_layout,
                                                 fig=True)      # (the FIBs carry the ensembles' FIGs: tools/dabfig.py)
        if synced:
            self.slices, self.cfo, self.toff = dabsynth.ensemble_slices(self.iq, self.mux.n, seed + 2, SLICE_LEAD, SLICE_SAMPLES, noise=noise)
            del self.iq
            torch.cuda.empty_cache()
            self.slices_f = torch.view_as_real(self.slices)        # [2][E][SLICE_SAMPLES][2]
            self.states = torch.zeros(E * 24, dtype=torch.uint8, device=device)        # dabgpu_sync_state records, persistent
            self.sync_cfg = dabgpu.sync_cfg_default()
        else:
            self.iq_f = torch.view_as_real(self.iq)                # [2][E][196608][2]
        self.frame_of_slot = {}                                # ring slot -> number of the frame it holds
        self.n_sub = len(self.mux.layout)
        self.cif_bytes = self.mux.cif_out_bytes
        self.hist = torch.zeros((E, self.H, 230400), dtype=torch.int8, device=device)
        self.ctxs = [ctx] + [dabgpu.Context(device.index) for _ in range(inflight - 1)]
        if self.lanes == "split":
            self.streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
        else:
            self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(inflight - 1)]
        mk = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=device) for _ in range(inflight)]     # noqa: E731
        self.corr = mk((E, 76, 2), torch.float32)
        self.fic_out, self.fic_res = mk((E, 4, 96), torch.uint8), mk((E * 4, 16), torch.uint8)
        self.msc_out, self.msc_res = mk((E, 4, self.cif_bytes), torch.uint8), mk((E * 4 * self.n_sub, 16), torch.uint8)
        self.subs = self.mux.subchannels(dabgpu)
        self.fic_steps = E * dabsynth.FIC_STEPS_PER_FRAME
        self.msc_steps = E * self.mux.msc_steps_per_frame
        self.stride = self.H * 230400
        self.j = 0                                  # next frame number
        self.ev_demod, self.ev_msc = {}, {}
        self.spb = 0

    def tune(self):
        """dabgpu_ofdm_tune once per context for this call shape (explicit, blocking, outside every timed region); the run length it
        records is what symbols_per_block = 0 resolves to afterwards"""
        src = self.slices_f[0] if self.synced else self.iq_f[0]
        chosen = []
        for c in self.ctxs:
            # (synced: the slices are longer than a frame; the calibration demodulates their first 196608 samples as frame-aligned
            # frames, which costs what the positioned frames cost)
            chosen.append(c.ofdm_tune(src, self.fmt_f32, self.E, self.hist[:, 0], bits_frame_stride=self.stride, bits_layout=self.layout,
                                      with_phase_tail=self.synced))
        self.torch.cuda.synchronize()
        return chosen

    # the stages of frame-slot `slot` on lane k (context k, stream k)
    def demod(self, slot, k=0, frame=None):
        """transmission frame `frame` (default: the next one after what the ring holds) of every ensemble into ring slot `slot`"""
        if frame is None:
            frame = self.frame_of_slot.get(slot, slot - self.H) + self.H       # (stage timing loops walk the ring in order)
        self.frame_of_slot[slot] = frame
        if self.synced:
            self.ctxs[k].ofdm_sync_demod_frames(self.slices_f[frame % self.mux.n_frames], self.E, SLICE_SAMPLES, SLICE_LEAD, self.states, self.hist[:, slot],
                                                cfg=self.sync_cfg, cp_corr=self.corr[k], bits_frame_stride=self.stride, bits_layout=self.layout,
                                                stream=self.streams[k].cuda_stream)
        else:
            self.ctxs[k].ofdm_demod_frames_history(self.iq_f[frame % self.mux.n_frames], self.fmt_f32, self.E, self.hist[:, slot], cp_corr=self.corr[k],
                                                   bits_frame_stride=self.stride, bits_layout=self.layout, stream=self.streams[k].cuda_stream)

    def sync_only(self, slot, k=0):
        """the synchroniser alone (5 transforms per receiver) on scratch records: what it costs inside demod()"""
        if not hasattr(self, "_scratch_states"):
            self._scratch_states = self.states.clone()
        self.ctxs[k].ofdm_sync(self.slices_f[slot % self.mux.n_frames].view(-1)[2 * SLICE_LEAD:], self.E, SLICE_SAMPLES, self._scratch_states,
                               cfg=self.sync_cfg, stream=self.streams[k].cuda_stream)

    def fic(self, slot, k=0):
        self.ctxs[k].fic_decode_frames(self.hist[:, slot], self.E, self.fic_out[k], self.fic_res[k], frame_stride=self.stride,
                                       stream=self.streams[k].cuda_stream)

    def msc(self, slot, k=0):
        self.ctxs[k].msc_decode_frames(self.hist, self.E, self.stride, self.H, slot, self.subs, self.msc_out[k],
                                       4 * self.cif_bytes, self.msc_res[k], stream=self.streams[k].cuda_stream, bits_layout=self.layout)

    def decode(self, slot, k=0):
        """FIC + MSC of the frame in ring slot `slot` in one call (dabgpu_decode_frames_layout: the FIB groups ride in the MSC launch)"""
        self.ctxs[k].decode_frames(self.hist, self.E, self.stride, self.H, slot, self.subs, self.fic_out[k], self.fic_res[k], self.msc_out[k],
                                   4 * self.cif_bytes, self.msc_res[k], stream=self.streams[k].cuda_stream, bits_layout=self.layout)

    def step(self, on_demod=None, decode=None):
        """one transmission frame of every ensemble: (sync ->) demod -> FIC + MSC   (decode = self.fic: configs[2])"""
        torch, j, n = self.torch, self.j, self.inflight
        k, slot, st = j % n, j % self.H, self.streams[j % n]
        decode = decode or self.decode
        self.j += 1
        self.last_frame_of_lane = getattr(self, "last_frame_of_lane", {})
        self.last_frame_of_lane[k] = j
        if n == 1:
            if on_demod:
                on_demod(lambda: self.demod(slot, 0, j))
            else:
                self.demod(slot, 0, j)
            decode(slot)
            return
        if self.lanes == "split":
            sd, stt = self.streams
            self.last_frame_of_lane = {1: j}
            w = self.ev_msc.pop(j - self.H + 4, None)                # the last reader of the slot this frame overwrites
            if w is not None:
                sd.wait_event(w)
            if on_demod:
                on_demod(lambda: self.demod(slot, 0, j))
            else:
                self.demod(slot, 0, j)
            ev = torch.cuda.Event(); ev.record(sd)
            stt.wait_event(ev)
            decode(slot, 1)
            ev = torch.cuda.Event(); ev.record(stt)
            self.ev_msc[j] = ev
            return
        w = self.ev_msc.pop(j - self.H + 4, None)                    # the last reader of the slot this frame overwrites
        if w is not None:
            st.wait_event(w)
        if self.synced:                                              # frame j's synchroniser reads what frame j - 1's phase tail wrote
            w = self.ev_demod.get(j - 1)
            if w is not None:
                st.wait_event(w)
        if on_demod:
            on_demod(lambda: self.demod(slot, k, j))
        else:
            self.demod(slot, k, j)
        ev = torch.cuda.Event(); ev.record(st)
        self.ev_demod[j] = ev
        for d in range(1, n):                                        # frames j-1 .. j-n+1 were demodulated on the other streams
            w = self.ev_demod.get(j - d)
            if w is not None:
                st.wait_event(w)
        self.ev_demod.pop(j - n, None)
        decode(slot, k)
        ev = torch.cuda.Event(); ev.record(st)
        self.ev_msc[j] = ev

    def fill(self):
        for _ in range(self.H + self.inflight):      # fill the history ring: the time de-interleaver needs 16 CIFs = 4 frames
            self.step()
        self.torch.cuda.synchronize()

    def timed(self, fn, reps):
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(self.streams[0])                   # (the stage functions launch on lane 0's stream)
        for k in range(reps):
            fn(k % self.H)
        e1.record(self.streams[0])
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def check(self, dabgpu, fic_only=False):
        """the outputs of the last frame of every lane against what was transmitted: frame j carries fibs[j mod 2], its CIF c decodes to
        payload[(4 j + c - 15) mod 8] -- the payload changes with every CIF, so a wrong ring slot, a wrong age or frames in flight in
        the wrong order cannot pass.  synced: plus what the synchroniser tracked, against what the generator put in."""
        import numpy as np
        torch, E, nd, P = self.torch, self.E, self.mux.n, self.mux.period
        out = {"fib_crc_pass": 0, "fib_crc_expected": E * 12 * len(self.last_frame_of_lane), "fib_bytes_equal_transmitted": True,
               "msc_bytes_equal_transmitted": True, "ensembles_checked": E, "distinct_multiplexes": nd, "frames_in_flight": self.inflight,
               "payload_period_cifs": P, "frames_checked": []}
        idx = torch.arange(E, device=self.hist.device) % nd
        for k in sorted(self.last_frame_of_lane):                    # the outputs of the last frame of every lane
            j = self.last_frame_of_lane[k]
            out["frames_checked"].append(int(j))
            exp = torch.stack([self.mux.expected_cif(4 * j + c)[idx] for c in range(4)], dim=1)      # [E, 4, bytes of a CIF's sub-channels]
            res_f = self.fic_res[k].cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(E, 4)
            out["fib_crc_pass"] += int(np.unpackbits(res_f["crc_ok_mask"].astype("<u4").view(np.uint8)).sum())
            out["fib_bytes_equal_transmitted"] &= bool(torch.equal(self.fic_out[k], self.mux.fibs[idx, j % self.mux.n_frames]))
            if not fic_only:
                out["msc_bytes_equal_transmitted"] &= bool(torch.equal(self.msc_out[k], exp))
        if fic_only:
            del out["msc_bytes_equal_transmitted"]
        if self.synced:
            st = self.states.cpu().numpy().view(np.dtype(dabgpu.SYNC_STATE_DTYPE))
            net = st["freq_coarse"].astype(np.float64) + st["freq_fine"].astype(np.float64)
            out["sync"] = {"receivers_with_valid_impulse_peak": int((st["sync_valid"] != 0).sum()),
                           "fine_time_offset_equals_generated": bool(np.array_equal(st["fine_time_offset"], self.toff.cpu().numpy())),
                           "carrier_offset_range_hz": [float(self.cfo.min().item()) * 2.048e6, float(self.cfo.max().item()) * 2.048e6],
                           "timing_offset_range_samples": [int(self.toff.min().item()), int(self.toff.max().item())],
                           # the PLL multiplies by e^{+j 2 pi f n}: a locked loop sits at minus the generator's offset
                           "max_abs_tracking_error_hz": float(np.abs(net + self.cfo.cpu().numpy().astype(np.float64)).max() * 2.048e6)}
        return out


def extras_configs23(ctx, dabgpu, torch, device, E, n_distinct, reps=6, layout=1, synced=True, lanes="alternate"):
    """BASELINE configs[2] (full demod incl. sync + FIC Viterbi) and configs[3] (full FIC + MSC, E concurrent ensembles) on this GPU"""
    p = Pipeline(ctx, dabgpu, torch, device, E, n_distinct, seed=7, inflight=2, layout=layout, synced=synced, lanes=lanes)
    chosen = p.tune()
    p.fill()
    torch.cuda.synchronize()
    t_demod, t_fic, t_msc = p.timed(p.demod, reps), p.timed(p.fic, reps), p.timed(p.msc, reps)      # one stage at a time, stream 0
    t_sync = p.timed(p.sync_only, reps) if synced else None
    t_dec = p.timed(p.decode, reps)                                            # FIC + MSC as one call (what the pipeline runs)
    # configs[2]: (sync ->) demod -> FIC, one frame at a time on one stream ...
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        p.demod(k % p.H); p.fic(k % p.H)
    torch.cuda.synchronize()
    t_c2_seq = (time.perf_counter() - t0) / reps * 1e3
    # ... and with two frames in flight like configs[3]: frame j on lane j mod 2, the FIC trellis of frame j beside the demodulation of j + 1
    p.fill()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4 * reps):
        p.step(decode=p.fic)
    torch.cuda.synchronize()
    t_c2 = (time.perf_counter() - t0) / (4 * reps) * 1e3
    chk2 = p.check(dabgpu, fic_only=True)
    # configs[3]
    p.fill()
    t0 = time.perf_counter()
    for k in range(reps):                                     # one frame at a time (both lanes' work serialised by the events of step())
        p.demod(k % p.H); p.decode(k % p.H)
    torch.cuda.synchronize()
    t_seq = (time.perf_counter() - t0) / reps * 1e3
    p.fill()
    t0 = time.perf_counter()
    for k in range(2 * reps):                                 # two frames in flight (Pipeline.step)
        p.step()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / (2 * reps) * 1e3
    chk = p.check(dabgpu)
    # DABGPU_VIT_MAP_AUTO's switch points for the FIC (include/dabgpu.h): wave -> octet at 1024 frames, octet -> lane at 10256
    lanes_fic = 64 if E >= 10256 else (8 if E >= 1024 else 0)
    fic_kernel = {64: "vit_lanes_kernel (FIC)", 8: "vit_octet_kernel (FIC)", 0: "viterbi_kernel (FIC)"}[lanes_fic]
    path = ("per frame and receiver: dabgpu_ofdm_sync_demod_frames (coarse + fine PRS synchronisation -> demodulation at the position and with the "
            "carrier offset found -> fine-frequency update; carrier offsets +-5 kHz, timing offsets +-100 samples)" if synced else
            "frame-aligned input, synchronisation bypassed (upper bound of the configuration)")
    kms = {"sync_and_demod_one_call": t_demod, "ofdm_sync_alone": t_sync} if synced else {"ofdm_demod": t_demod}
    c2 = {"workload": f"BASELINE configs[2]: full OFDM demod incl. sync + FIC Viterbi (4 x 774 trellis steps per frame), {E} frames", "frames": E,
          "path": path, "ms_per_step": t_c2, "frames_per_s": E / t_c2 * 1e3, "x_realtime": E / t_c2 * 1e3 / REALTIME_FRAMES_PER_S,
          "frames_in_flight": 2, "ms_per_step_one_frame_at_a_time": t_c2_seq, "frames_per_s_one_frame_at_a_time": E / t_c2_seq * 1e3,
          "kernel_ms": dict(kms, fic_viterbi=t_fic), "symbols_per_block": chosen[0],
          "roofline": [hbm_roofline("ofdm_sync_kernel + ofdm_demod_kernel" if synced else "ofdm_demod_kernel", t_demod, E),
                       viterbi_roofline(fic_kernel, p.fic_steps, t_fic, lanes_fic)],
          "check": chk2}
    c3 = {"workload": f"BASELINE configs[3]: full FIC + MSC demod + Viterbi, {E} concurrent synthetic ensembles, 18 x 48 CU EEP 3-A", "ensembles": E,
          "path": path, "ms_per_step": t_all, "frames_per_s": E / t_all * 1e3, "x_realtime": E / t_all * 1e3 / REALTIME_FRAMES_PER_S,
          "frames_in_flight": 2, "history_layout": "time-interleaver class order" if layout else "natural",
          "ms_per_step_one_frame_at_a_time": t_seq, "frames_per_s_one_frame_at_a_time": E / t_seq * 1e3,
          "kernel_ms": dict(kms, fic_viterbi=t_fic, msc_viterbi_incl_deinterleave=t_msc, fic_and_msc_one_call=t_dec), "symbols_per_block": chosen[0],
          "algorithmic_hbm_GBps": 2.27e6 * E / (t_all * 1e-3) / 1e9,
          "roofline": [hbm_roofline("ofdm_sync_kernel + ofdm_demod_kernel" if synced else "ofdm_demod_kernel", t_demod, E),
                       viterbi_roofline(("vit_prep_ring4c_kernel" if layout else "vit_prep_ring4_kernel") + " + vit_lanes_kernel (MSC)", p.msc_steps, t_msc, True)],
          "check": chk}
    host_sample = None
    if synced:                                                # a few receivers' slices for the CPU baseline of the same path (untimed copy)
        import numpy as np
        ne = min(4, E)
        host_sample = {"slices": p.slices[:, :ne].cpu().numpy(), "fibs": p.mux.fibs[:ne].cpu().numpy(), "payload": p.mux.payload[:ne].cpu().numpy(),
                       "period": p.mux.period, "n_frames": p.mux.n_frames}
    del p
    torch.cuda.empty_cache()
    return c2, c3, host_sample


def extras_mixed(ctx, dabgpu, torch, device, E, n_distinct, reps=6, layout=1, synced=True, lanes="alternate"):
    """configs[3] on a HETEROGENEOUS multiplex (tools/dabsynth.py::mixed_layout: 14 sub-channels, 8 sizes, EEP 3-A / EEP 2-B / three UEP rows /
    the 8 CU EEP 2-A case -- msc_decoder.cpp:77-154, subchannel_protection_tables.h:21-139): the same calls as configs[3], and the decode alone
    under every forced mapping beside the one DABGPU_VIT_MAP_AUTO's cost model takes"""
    import dabsynth
    mux_layout = dabsynth.mixed_layout()
    dabsynth.check_layout(mux_layout, dabgpu)
    p = Pipeline(ctx, dabgpu, torch, device, E, n_distinct, seed=11, inflight=2, layout=layout, synced=synced, lanes=lanes, mux_layout=mux_layout)
    p.tune()
    p.fill()
    torch.cuda.synchronize()
    chosen, model = ctx.multiplex_mapping(E, p.subs)
    names = {1: "wave", 2: "lane", 3: "octet"}
    ab = {}
    for m in (2, 3, 1):                                          # the decode call alone (FIC + MSC of one frame of every ensemble), mapping forced
        for c in p.ctxs:
            c.viterbi_set_mapping(m)
        p.timed(p.decode, 1)
        ab[names[m]] = p.timed(p.decode, reps if m != 1 else 2)
    for c in p.ctxs:
        c.viterbi_set_mapping(0)
    p.timed(p.decode, 1)
    t_dec = p.timed(p.decode, reps)
    t_demod = p.timed(p.demod, reps)
    p.fill()
    t0 = time.perf_counter()
    for k in range(2 * reps):
        p.step()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / (2 * reps) * 1e3
    chk = p.check(dabgpu)
    steps = p.msc_steps + p.fic_steps
    out = {"workload": f"configs[3] on a mixed multiplex: {E} concurrent synthetic ensembles x {len(mux_layout)} sub-channels "
                       "(EEP 3-A 3 x 48 / 3 x 60 / 2 x 72 CU, EEP 2-B 2 x 42 CU, UEP rows 35 / 38 / 43 = 128 / 160 / 192 kbit/s, EEP 2-A 8 CU; 832 of 864 CU)",
           "ensembles": E, "sub_channels": [{k: d[k] for k in ("start", "length", "is_uep", "uep_index", "eep_level", "eep_type", "nbytes")} | {"trellis_steps": sum(32 * L for _, L in d["segments"]) + 6}
                                            for d in mux_layout],
           "trellis_steps_per_frame": {"msc": p.mux.msc_steps_per_frame, "fic": 4 * 774},
           "ms_per_step": t_all, "frames_per_s": E / t_all * 1e3, "x_realtime": E / t_all * 1e3 / REALTIME_FRAMES_PER_S, "frames_in_flight": 2,
           "kernel_ms": {"sync_and_demod_one_call" if synced else "ofdm_demod": t_demod, "fic_and_msc_one_call": t_dec},
           "mapping": {"auto_chose": names[chosen], "cost_model_us": model, "decode_call_ms_forced": ab, "decode_call_ms_auto": t_dec,
                       "what": "DABGPU_VIT_MAP_AUTO takes ONE mapping for all sub-channels of a call (include/dabgpu.h); the forced timings are the same call, same "
                               "inputs, mapping set with dabgpu_viterbi_set_mapping"},
           "trellis_steps_per_s_decode_call": steps / (t_dec * 1e-3),
           "check": chk}
    del p
    torch.cuda.empty_cache()
    return out


def cpu_baseline_full(sample, seconds_target=14.0):
    """the oracle's whole receive chain per frame -- coarse + fine synchronisation, demodulation, fine-frequency update, 4 FIB groups,
    18 sub-channels x 4 CIFs through CIF de-interleaver + Viterbi + descrambler (oracle/dab_oracle_chain.c: ONE C call per thread, no
    interpreter inside) -- on this box's host cores, on the very slices the GPU just decoded; bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import numpy as np
    import oracle as O
    O.lib()
    slices, nf, period = sample["slices"], sample["n_frames"], sample["period"]
    ne = slices.shape[1]
    subs = [O.subchannel(48 * s_, 48, eep_level=2, eep_type=0) for s_ in range(18)]
    per = [np.ascontiguousarray(slices[:, e]) for e in range(ne)]                   # receiver e: its nf stored slices
    warm = 6                                                                         # loops settle, time de-interleavers fill (>= 4 frames)
    t0 = time.perf_counter()
    r1 = O.receive_frames(per[0], SLICE_SAMPLES, SLICE_LEAD, warm, subs)
    dt1 = (time.perf_counter() - t0) / warm
    cpus, topo = host_cores()
    cores = len(cpus)

    def run(per_thread):
        return run_pinned(cpus, lambda i: O.receive_frames(per[i % ne], SLICE_SAMPLES, SLICE_LEAD, per_thread, subs))

    _, probe = run(warm + 2)                                                        # calibrate: all cores busy scale far from linearly on some hosts
    per_thread = int(max(warm + 2, min(4096, (warm + 2) * seconds_target / probe)))
    res, dt = run(per_thread)
    ok = True
    for i, r in enumerate(res):                                                     # every thread's last frame against what was transmitted
        e, j = i % ne, per_thread - 1
        ok &= bool(r["sync_failed"] == 0 and np.array_equal(r["fib"], sample["fibs"][e, j % nf]))
        cifs = [(4 * j + c - 15) % period for c in range(4)]
        ok &= bool(np.array_equal(r["msc"], sample["payload"][e][cifs].reshape(4, -1)))
    done = per_thread * cores
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "threads": cores, "kind": "port", "single_thread_value": 1.0 / dt1,
            "scaling_efficiency": (done / dt) / (cores * (1.0 / dt1)), "topology": topo,
            "decoded_bytes_equal_transmitted": ok, "calibration": cpu_calibration(),
            "sample": f"{done} frames through oracle/dab_oracle_chain.c dab_receive_frames (coarse + fine sync, PLL + CP phase + 76 x FFT2048 + DQPSK + demap, "
                      f"fine-frequency update, 4 FIB groups, 18 sub-channels x 4 CIFs: CIF de-interleaver + K=7 Viterbi + descrambler), "
                      f"{cores} pinned worker threads (one per physical core) x {per_thread} frames of the receivers the GPU decoded (incl. {warm} settling frames each), {dt:.1f} s wall"}


def dry_run(args, rank, world):
    """the multi-rank plumbing of main() without a GPU: rendezvous (gloo), shard ranges, barrier, max over ranks, one line from rank 0"""
    import torch
    import torch.distributed as dist
    from dabgpu import shard
    if rank == args.dry_run_fail_rank:
        sys.exit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    units = args.frames if args.workload == "demod" else args.ensembles
    first, n = shard.shard_range(units * world, rank, world)
    shard.barrier(dist if world > 1 else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    own = time.perf_counter() - t0                      # this rank's own steps, before it waits for the others (as in main())
    shard.barrier(dist if world > 1 else None)
    mine = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(mine, dist if world > 1 else None)
    per_rank = shard.all_ranks(own / args.steps * 1e3, dist if world > 1 else None)
    covered = shard.sum_over_ranks(n, dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "dab_mode1_frames_per_sec", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run (no kernels)", "ms_per_step_per_rank": per_rank,
                          "config": {"workload": args.workload, "units_per_rank": n, "units_covered": covered, "first_unit_rank0": first}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np  # noqa: F401
    import torch
    import dabgpu
    from dabgpu import shard
    import dabsynth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    ctx = dabgpu.Context(local_rank)
    prs, mapper, _ = dabgpu.host_tables()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    evs = []                                                  # (start, stop) event pairs around demod launches inside the timed loop


    if args.workload == "demod":
        F = args.frames
        # weak scaling: the frame set grows with the number of GPUs; rank r owns the contiguous block shard_range gives
        first_unit, n_units = shard.shard_range(F * world, rank, world)
        assert n_units == F
        iq, tx_bits, freq = dabsynth.random_frames(F, seed=1000 + first_unit, device=device, mapper=mapper, prs=prs)
        d_freq = freq.clone()                                 # the PLL multiplies by e^{+j 2 pi f n}: the generator applied -f
        d_bits = torch.empty((F, 230400), dtype=torch.int8, device=device)
        d_corr = torch.empty((F, 76, 2), dtype=torch.float32, device=device)
        d_total = torch.empty(F, dtype=torch.float32, device=device)
        d_fine = torch.zeros(F, dtype=torch.float32, device=device)
        iq_f = torch.view_as_real(iq)
        units = F

        fmt_f32 = dabgpu.IQ_FORMATS.index("raw_f32l")
        # symbols per workgroup: --spb 0 (default) leaves the choice to the library: dabgpu_ofdm_tune (called here, in the warm-up) times
        # a whole frame (75: one round of workgroups on a full chip, phase tail inside the kernel), two and three runs per frame (38, 25)
        # and records the fastest for this call shape (include/dabgpu.h; which one wins depends on the box); the data path
        # only looks the record up.  The three are timed here once more (untimed region) only to put the numbers in the line
        spb_timing = None
        if args.spb == 0 and F >= 512 and not args.dry_run:
            # explicit, blocking calibration (dabgpu_ofdm_tune), outside every timed region; symbols_per_block = 0 below then resolves to it
            ctx.ofdm_tune(iq_f, fmt_f32, F, d_bits, with_phase_tail=True)
            torch.cuda.synchronize()
            spb_timing = {}
            for rep in range(2):                                       # interleaved, the last pass counts: all see the same clock state
                for cand in (25, 38, 75):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ctx.ofdm_demod_phase_frames(iq_f, fmt_f32, F, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=cand,
                                                    beta=0.9, total_phase=d_total, fine_freq=d_fine)
                    e1.record(); torch.cuda.synchronize()
                    spb_timing[cand] = e0.elapsed_time(e1) / 20
            d_fine.zero_()

        def demod_launch():
            # demodulation + the phase tail of the fine-frequency loop (ofdm_phase_update) as one call: one launch when a workgroup walks
            # a whole frame (75), else the tail follows as its own small launch inside the call
            ctx.ofdm_demod_phase_frames(iq_f, fmt_f32, F, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=args.spb,
                                        beta=0.9, total_phase=d_total, fine_freq=d_fine)

        def step(k, timed=False):
            demod_launch()          # (timed by ONE pair of HIP events around the whole timed loop, below: nothing is recorded between launches)
    else:
        E = args.ensembles
        first_unit, n_units = shard.shard_range(E * world, rank, world)
        assert n_units == E
        pipe = Pipeline(ctx, dabgpu, torch, device, E, args.distinct, seed=5000 + first_unit, inflight=args.inflight,
                        layout=int(args.hist_layout == "classed"), synced=not args.aligned, lanes=args.lanes)
        pipe_spb = pipe.tune()
        pipe.fill()
        units = E

        def timed_demod(launch):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = pipe.streams[0] if pipe.lanes == "split" else pipe.streams[(pipe.j - 1) % pipe.inflight]
            a.record(st); launch(); b.record(st)
            evs.append((a, b))

        def step(k, timed=False):
            pipe.step(on_demod=timed_demod if (timed and k % 2 == 0) else None)

    t_pre = time.perf_counter()
    k = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:      # clock settling, untimed and not part of W
        for _ in range(20 if args.workload == "demod" else 1):
            step(k); k += 1
        torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k)
    barrier()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if args.workload == "demod" and not args.dry_run else None
    t0 = time.perf_counter()
    if region:
        region[0].record()
    for k in range(args.steps):
        step(k, timed=True)
    if region:
        region[1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = shard.all_ranks(elapsed / args.steps * 1e3, dist, device)        # every rank's own clock over the same barrier-to-barrier region
    elapsed = shard.max_over_ranks(elapsed, dist, device)
    ms_per_step = elapsed / args.steps * 1e3
    if region:
        # configs[1]: the launches of the timed loop run back to back on this stream; one event pair around all of them gives the mean
        # launch duration without opening a gap in front of every launch (a pair per launch cost ~1.5 % of a 0.4 ms step)
        k_ms = region[0].elapsed_time(region[1]) / args.steps
    else:
        k_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))  # mean demod launch duration inside the timed loop

    # ---- correctness of what was just timed (untimed) ----
    check = {}
    if args.workload == "demod" and not args.no_check:
        hard = (d_bits.view(F, 75, 3072) >= 0).to(torch.uint8)
        err = (hard != tx_bits)
        check["hard_bit_errors_vs_transmitted"] = int(err.sum().item())
        check["frames_with_hard_bit_errors"] = int((err.view(F, -1).sum(dim=1) > 0).sum().item())
        check["frames_checked"] = F
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            mism = 0
            for kk in (0, F // 2, F - 1):
                exp = O.demod_frame(iq[kk].cpu().numpy(), float(freq[kk].item()))
                mism += int((exp["bits"] != d_bits[kk].cpu().numpy()).sum())
            check["soft_bit_mismatches_vs_oracle_3_frames"] = mism
    elif args.workload == "full" and not args.no_check:
        check = pipe.check(dabgpu)

    if rank == 0:
        assert k_ms <= ms_per_step * 1.02, f"demod launch {k_ms} ms cannot exceed the step {ms_per_step} ms it is part of (2 % event jitter allowed)"
        # a pair of events brackets the launch AND the gap the event records themselves open in front of it (~2 us); when the step is
        # that one launch (phase tail inside the kernel) the pair can therefore read a little more than the step: the step bounds it
        k_ms_events = k_ms
        k_ms = min(k_ms, ms_per_step)
        value = world * units * args.steps / elapsed
        if args.workload == "demod":
            workload = ("BASELINE configs[1]: batched 1024 Mode-I frames of synthetic IQ (c32, HBM-resident), "
                        "PLL+CP-phase+FFT2048+DQPSK+demap, per GPU")
            chosen = args.spb or ctx.ofdm_tuned_symbols_per_block(fmt_f32, units, with_phase_tail=True)
            config = {"workload": workload, "frames_per_gpu_per_step": units, "symbols_per_block": chosen,
                      "symbols_per_block_chosen_by": "caller (--spb)" if args.spb else "library (symbols_per_block = 0 resolves to what dabgpu_ofdm_tune recorded in the warm-up)",
                      "sharding": "independent frames / ensembles per rank, no data-path collective"}
            if spb_timing:
                config["symbols_per_block_timed_ms"] = {str(k): v for k, v in spb_timing.items()}
        else:
            workload = (f"BASELINE configs[4]: {units * world} synthetic ensembles ({units} per GPU, built on the device from "
                        f"{min(args.distinct, units)} seeded multiplexes), full FIC+MSC: " +
                        ("frame-aligned OFDM demod (sync bypassed)" if args.aligned else
                         "PRS synchronisation (coarse + fine) + OFDM demod at the tracked carrier / timing offset + fine-frequency update") +
                        " + FIC Viterbi + 18 x 48 CU EEP 3-A time de-interleave / Viterbi / descramble per transmission frame")
            config = {"workload": workload, "ensembles_per_gpu": units, "frames_per_gpu_per_step": units, "symbols_per_block": pipe_spb[0],
                      "sharding": "independent ensembles per rank, no data-path collective"}
        line = {
            "metric": "dab_mode1_frames_per_sec", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config,
            "x_realtime": value / REALTIME_FRAMES_PER_S,
            "ms_per_step_per_rank": per_rank_ms,
            "roofline": hbm_roofline("ofdm_demod_kernel", k_ms, units),
            "check": check,
        }
        if region:
            line["roofline"]["timing"] = (f"one pair of HIP events around the {args.steps} back-to-back launches of the timed loop, on their stream: "
                                          f"{k_ms_events:.4f} ms per launch" + ("" if (args.spb or ctx.ofdm_tuned_symbols_per_block(fmt_f32, units, with_phase_tail=True)) == 75 else " (incl. the 5 us phase-tail launch of each step)"))
        else:
            # configs[4]: inside the timed loop the demodulation of frame j + 1 runs BESIDE the trellis kernel of frame j (two frames in
            # flight), so its launches last longer than the kernel needs; the rooflines are taken one stage at a time after the loop
            pipe.timed(pipe.msc, 4); pipe.timed(pipe.demod, 8)              # (the clock has dropped during the host-side check above)
            t_d, t_f, t_m = pipe.timed(pipe.demod, 6), pipe.timed(pipe.fic, 6), pipe.timed(pipe.msc, 6)
            line["roofline"] = hbm_roofline("ofdm_demod_kernel" if args.aligned else "ofdm_sync_kernel + ofdm_demod_kernel", t_d, units)
            line["roofline"]["timing"] = (f"one stage at a time after the timed loop: HIP events around 6 launches, {t_d:.4f} ms each (inside the loop, "
                                          f"overlapped with the other frame's trellis kernel: {k_ms_events:.4f} ms, mean of {len(evs)} launches)")
            line["roofline_decode"] = viterbi_roofline(("vit_prep_ring4c_kernel" if pipe.layout else "vit_prep_ring4_kernel") + " + vit_lanes_kernel (MSC)",
                                                       pipe.msc_steps, t_m, True)
            line["stage_ms_one_at_a_time"] = {("ofdm_demod" if args.aligned else "sync_and_demod_one_call"): t_d, "fic_viterbi": t_f, "msc_viterbi_incl_deinterleave": t_m}
            if not args.aligned:
                line["stage_ms_one_at_a_time"]["ofdm_sync_alone"] = pipe.timed(pipe.sync_only, 6)
        if args.workload == "demod" and world == 1 and not args.dry_run:
            # What the fraction means.  The roofline the path is PRICED against is HBM -- traffic = 0.99 x algorithmic bytes, nothing is
            # re-read -- and `frac` stays the north-star figure against 8 TB/s.  What LIMITS the kernel at that fraction is VALU issue at the clock the
            # power manager grants: 752 VALU instructions per thread and symbol under the bit-exact contract (the reference's per-sample Chebyshev PLL and
            # two correctly rounded quotients per carrier), ~79 % of the issue slots at ~1.75 GHz (profiles/r05/counters_v5.json, pmc_*_v5.csv).
            line["roofline"]["limiter"] = "valu_issue@power"
            line["roofline"]["limiter_evidence"] = ("HBM traffic 0.99 x algorithmic (no wasted bytes); 752 VALU instructions per thread and symbol = ~79 % of issue "
                                                    "slots at the ~1.75 GHz the power manager sustains; min launch 349.8 us = 0.66 of 8 TB/s shows what the clock costs "
                                                    "(profiles/r05/counters_v5.json, profiles/r06/)")
            # the launch-granularity share: the same kernel on 4096 frames per launch (1024 frames = 1024 workgroups of a whole frame = ONE round on
            # 256 CUs x 4: the tail of the round is idle time the figure above includes).  Not under --no-extras: the profile passes (rocprofv3 kernel
            # stats, PMC) of that command must see launches of ONE shape only
            try:
                if args.no_extras:
                    raise RuntimeError("skipped (--no-extras)")
                F4 = 4 * units
                big_iq = iq_f.repeat(4, 1, 1).contiguous()
                big_bits = torch.empty((F4, 230400), dtype=torch.int8, device=device)
                big_freq = d_freq.repeat(4).contiguous()
                big_corr = torch.empty((F4, 76, 2), dtype=torch.float32, device=device)
                big_total = torch.empty(F4, dtype=torch.float32, device=device)
                big_fine = torch.zeros(F4, dtype=torch.float32, device=device)

                def big_launch():
                    ctx.ofdm_demod_phase_frames(big_iq, fmt_f32, F4, big_bits, freq_offset=big_freq, cp_corr=big_corr, symbols_per_block=args.spb,
                                                beta=0.9, total_phase=big_total, fine_freq=big_fine)
                if args.spb == 0:
                    ctx.ofdm_tune(big_iq, fmt_f32, F4, big_bits, with_phase_tail=True)       # (the library's record for THIS call shape, as for the 1024-frame one)
                for _ in range(8):
                    big_launch()
                torch.cuda.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(10):
                    big_launch()
                g1.record(); torch.cuda.synchronize()
                big_ms = g0.elapsed_time(g1) / 10
                line["roofline"]["frac_4096_frames_per_launch"] = ALGO_BYTES_PER_FRAME * F4 / (big_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                line["roofline"]["kernel_ms_4096_frames_per_launch"] = big_ms
                assert torch.equal(big_bits[:units], d_bits) and torch.equal(big_bits[3 * units:], d_bits), "the 4096-frame launch must give the same soft bits"
                del big_iq, big_bits, big_freq, big_corr, big_total, big_fine
            except Exception as ex:                                   # (memory: 6.4 GB more; never costs the line)
                line["roofline"]["frac_4096_frames_per_launch"] = None
                line["roofline"]["frac_4096_error"] = str(ex)[-200:]
        # PMC-derived HBM traffic per launch, when a profiles/ summary of this round exists (see profiles/README.md)
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
                tr = json.load(fh)
            if tr.get("frames_per_launch") == units:
                line["roofline"]["traffic"] = tr.get("bytes_per_launch")
                line["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass
        host_sample = None
        if world == 1 and args.workload == "demod" and not args.no_extras:
            del iq, iq_f, tx_bits, d_bits
            torch.cuda.empty_cache()
            c2, c3, host_sample = extras_configs23(ctx, dabgpu, torch, device, args.extra_ensembles, args.distinct, layout=int(args.hist_layout == "classed"),
                                                   synced=not args.aligned, lanes=args.lanes)
            line["extra"] = {"configs2": c2, "configs3": c3}
            if not args.no_mixed:
                line["extra"]["configs3_mixed"] = extras_mixed(ctx, dabgpu, torch, device, args.extra_ensembles, args.distinct, layout=int(args.hist_layout == "classed"),
                                                               synced=not args.aligned, lanes=args.lanes)
            if not args.no_mirror:
                # ONE receiver behind the drop-in classes (the path basic_radio would call): the C++ harness as a child process, ~4 s
                try:
                    import bench_mirror
                    line["extra"]["one_receiver"] = bench_mirror.run_mirror(torch, dabgpu, frames=200, loops=10, repeats=3, variants=("frame_batcher_one_thread",))
                except Exception as ex:                                    # the harness is test plumbing: its absence must not cost the bench line
                    line["extra"]["one_receiver"] = {"error": str(ex)[-300:]}
                # EIGHT receivers of one process behind the classes (reader + delivery + radio thread each, 2000 frames per receiver): every one on a pipeline
                # of its own, every one a member of the receiver bank, and the classes' own rule (the first private, the others members) -- ~15 s
                try:
                    import bench_mirror_multi
                    mm = bench_mirror_multi.run_multi(torch, dabgpu, receivers=(8,), frames=100, loops=20, modes=("0", "1", None))
                    if mm is not None:
                        line["extra"]["eight_receivers"] = {
                            "what": mm["what"], "frames_per_receiver": 2000,
                            "frames_per_s": {("private_pipelines" if r["bank"] == "0" else ("receiver_bank" if r["bank"] == "1" else "classes_own_rule")): r["frames_per_s"] for r in mm["runs"]},
                            "host_cpu_ms_per_frame": {("private_pipelines" if r["bank"] == "0" else ("receiver_bank" if r["bank"] == "1" else "classes_own_rule")): r["host_cpu_ms_per_frame"] for r in mm["runs"]},
                            "memory_of_the_bank_run": [r["memory"] for r in mm["runs"] if r["bank"] == "1"][0]}
                except Exception as ex:
                    line["extra"]["eight_receivers"] = {"error": str(ex)[-300:]}
            if not args.no_chain:
                import bench_chain
                line["extra"]["chain"] = bench_chain.run_chain(ctx, dabgpu, torch, device, args.extra_ensembles, args.distinct,
                                                               layout=int(args.hist_layout == "classed"))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
            if host_sample is not None:
                line["cpu_baseline_full"] = cpu_baseline_full(host_sample)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
