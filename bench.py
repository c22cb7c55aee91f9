#!/usr/bin/env python3
"""bench.py -- Mode-I frames/s of the MI355X-native DAB receive path (driver contract: one JSON line from rank 0).

Workloads
  demod (default; BASELINE.json configs[1], the configuration the metric is quoted on): a batch of 1024 frame-aligned Mode-I
        frames of synthetic IQ per GPU, resident in HBM as complex float32, through the fused PLL + cyclic-prefix phase +
        2048-pt FFT + DQPSK + frequency de-interleave + soft-bit kernel and the per-frame phase / fine-frequency tail.
        One "step" = one pass over the batch.  At N = 1 the line also carries `extra.configs2` / `extra.configs3`
        (demod + FIC Viterbi, and full FIC + MSC for 4096 concurrent ensembles) with their own roofline blocks.
  full  (BASELINE.json configs[4], per GPU): 8192 ensembles per GPU (built on the device from <= 64 seeded multiplexes),
        one step = demodulate one transmission frame of every ensemble into its frame-history ring + FIC Viterbi (4 FIB groups)
        + MSC time de-interleave, Viterbi and descrambling of 18 sub-channels x 4 CIFs.

N > 1: every rank owns an independent block of frames / ensembles (dabgpu.shard.shard_range), no collective in the data path;
torch.distributed (RCCL) carries only the timing barrier and the max over ranks -> "scaling": "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload demod|full]
`python bench.py --gpus N` started plainly (no RANK in the environment) launches N ranks itself through
torch.distributed.run -- as a child process, before this process touches the GPU -- and exits with the child's status.

`roofline` (dominant kernel ofdm_demod_kernel): achieved = algorithmic bytes per launch (1,803,264 B/frame x frames) /
mean launch duration measured with HIP events INSIDE the timed loop (an event pair around every 8th demod launch on the launch
stream; around every launch when K <= 100); `cpu_baseline` = the oracle (C port of the reference algorithm) timed on this box's host cores on a bounded sample.
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


def hbm_roofline(kernel, k_ms, frames):
    achieved = ALGO_BYTES_PER_FRAME * frames / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "achieved_excl_null": achieved * (1.0 - NULL_SYMBOL_BYTES / ALGO_BYTES_PER_FRAME),
            "traffic": None, "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * frames}


def _counter_evidence():
    """VALU instructions per wavefront and the clock under the profiler of the batch Viterbi kernels, from the newest
    profiles/r03/counters_v*.json (tools/prof_counters.sh: rocprofv3 --pmc passes at 4096 ensembles); {} when no summary is there"""
    import glob
    out = {}
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03", "counters_v*.json")), key=lambda p: int(p.rsplit("_v", 1)[1].split(".")[0]))[-1]
        with open(path) as fh:
            k = json.load(fh)["kernels"]
        for name, steps, key in (("vit_lanes_kernel<0, 1, 5>", 1542.0, 64.0), ("vit_octet_kernel<0>", 774.0, 8.0)):
            if name in k and k[name].get("valu_per_wave"):
                out[key] = {"instr_per_step": k[name]["valu_per_wave"] / steps, "clock_ghz_profiled": k[name].get("clock_ghz_profiled"),
                            "valu_issue_cycles_frac": k[name].get("valu_issue_cycles_frac"), "source": os.path.relpath(path, ROOT)}
    except Exception:
        pass
    return out


def viterbi_roofline(kernel, steps, k_ms, lanes):
    """VALU-issue bound of the trellis recursion; `lanes`: codewords per wavefront of the mapping -- 64 (one lane per codeword), 8 (eight
    lanes per codeword) or 1 / False (one wavefront per codeword).  Counter evidence (VALU instructions per wavefront, issue share, clock under the profiler): profiles/r03/counters_*.json"""
    per_wave = float(lanes) if lanes else 1.0
    if lanes is True:
        per_wave = 64.0
    instr = {64.0: VIT_LANES_INSTR_PER_STEP, 8.0: VIT_OCTET_INSTR_PER_STEP}.get(per_wave, VIT_WAVE_INSTR_PER_STEP)
    ev = _counter_evidence().get(per_wave)
    if ev:                                            # measured: SQ_INSTS_VALU / SQ_WAVES / trellis steps per codeword
        instr = round(ev["instr_per_step"], 2)
    peak = N_SIMD * CLOCK_HZ * per_wave / (instr * VIT_CYCLES_PER_INSTR) / 1e9
    achieved = steps / (k_ms * 1e-3) / 1e9
    out = {"bound": "valu_issue", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "G trellis steps/s", "frac": achieved / peak,
           "kernel_ms": k_ms, "trellis_steps_per_launch": steps,
           "peak_definition": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9} GHz x {int(per_wave)} codewords per wavefront / ({instr} VALU instructions per "
                              f"wavefront-step x {VIT_CYCLES_PER_INSTR} cycles)"}
    if ev:
        out["counters"] = {"source": ev["source"], "valu_instructions_per_wavefront_step": ev["instr_per_step"],
                           "clock_ghz_under_profiler": ev["clock_ghz_profiled"], "valu_issue_share_of_simd_cycles": ev["valu_issue_cycles_frac"],
                           "peak_at_that_clock": peak * (ev["clock_ghz_profiled"] or 0.0) * 1e9 / CLOCK_HZ if ev["clock_ghz_profiled"] else None}
    return out


class Pipeline:
    """E ensembles, one transmission frame of IQ each, frame-history ring of H slots, FIC + MSC outputs (configs[2]/[3]/[4]).
    A step decodes the FIC and the MSC of a frame with ONE call (dabgpu_decode_frames_layout): the 4 FIB groups of every ensemble join the
    MSC's trellis launch, whose last round of wavefront slots the MSC's own groups do not fill.

    `inflight` frames are in flight at once, frame j on stream / context j mod inflight (a context owns its scratch, so concurrent
    calls need one each): the next frame's HBM-bound demodulation and gather kernels fill the wavefront slots that the trellis kernel's
    last, partial round leaves idle.  Dependencies kept with events: msc(j) reads the ring slots of frames j-4..j -> waits for
    demod(j-1), ...; demod(j) overwrites the slot of frame j-H, last read by msc(j-H+4) -> waits for it."""

    def __init__(self, ctx, dabgpu, torch, device, E, n_distinct, seed, inflight=1, layout=1):
        import dabsynth
        self.torch, self.E, self.inflight = torch, E, inflight
        # layout of the MSC soft bits in the history ring: 1 = time-interleaver class order (DABGPU_BITS_MSC_CLASSED: the demodulator
        # writes it for free and the decoder's gather then reads ~1.3 instead of 4.75 history bytes per soft bit), 0 = On_OFDM_Frame()
        self.layout, self.fmt_f32 = layout, dabgpu.IQ_FORMATS.index("raw_f32l")
        self.H = 5 if inflight == 1 else 8
        prs, mapper, _ = dabgpu.host_tables()
        # two stored transmission frames that repeat (8 CIFs of changing payload, time interleaved): decoded bytes then prove WHICH
        # ring slots / ages / frames in flight they came from (tools/dabsynth.py)
        self.iq, self.mux = dabsynth.ensemble_iq(E, min(n_distinct, E), seed, device, mapper, prs)
        self.iq_f = torch.view_as_real(self.iq)                # [2][E][196608][2]
        self.frame_of_slot = {}                                # ring slot -> number of the frame it holds
        self.n_sub = dabsynth.N_SUB
        self.hist = torch.zeros((E, self.H, 230400), dtype=torch.int8, device=device)
        self.ctxs = [ctx] + [dabgpu.Context(device.index) for _ in range(inflight - 1)]
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(inflight - 1)]
        mk = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=device) for _ in range(inflight)]     # noqa: E731
        self.corr = mk((E, 76, 2), torch.float32)
        self.fic_out, self.fic_res = mk((E, 4, 96), torch.uint8), mk((E * 4, 16), torch.uint8)
        self.msc_out, self.msc_res = mk((E, 4, self.n_sub * dabsynth.SUB_BYTES), torch.uint8), mk((E * 4 * self.n_sub, 16), torch.uint8)
        self.subs = self.mux.subchannels(dabgpu)
        self.fic_steps = E * dabsynth.FIC_STEPS_PER_FRAME
        self.msc_steps = E * dabsynth.MSC_STEPS_PER_FRAME
        self.stride = self.H * 230400
        self.j = 0                                  # next frame number
        self.ev_demod, self.ev_msc = {}, {}

    # the three stages of frame-slot `slot` on lane k (context k, stream k)
    def demod(self, slot, k=0, frame=None):
        """transmission frame `frame` (default: the next one after what the ring holds) of every ensemble into ring slot `slot`"""
        if frame is None:
            frame = self.frame_of_slot.get(slot, slot - self.H) + self.H       # (stage timing loops walk the ring in order)
        self.frame_of_slot[slot] = frame
        self.ctxs[k].ofdm_demod_frames_history(self.iq_f[frame % self.mux.n_frames], self.fmt_f32, self.E, self.hist[:, slot], cp_corr=self.corr[k],
                                               bits_frame_stride=self.stride, bits_layout=self.layout, stream=self.streams[k].cuda_stream)

    def fic(self, slot, k=0):
        self.ctxs[k].fic_decode_frames(self.hist[:, slot], self.E, self.fic_out[k], self.fic_res[k], frame_stride=self.stride,
                                       stream=self.streams[k].cuda_stream)

    def msc(self, slot, k=0):
        self.ctxs[k].msc_decode_frames(self.hist, self.E, self.stride, self.H, slot, self.subs, self.msc_out[k],
                                       4 * self.n_sub * 192, self.msc_res[k], stream=self.streams[k].cuda_stream, bits_layout=self.layout)

    def decode(self, slot, k=0):
        """FIC + MSC of the frame in ring slot `slot` in one call (dabgpu_decode_frames_layout: the FIB groups ride in the MSC launch)"""
        self.ctxs[k].decode_frames(self.hist, self.E, self.stride, self.H, slot, self.subs, self.fic_out[k], self.fic_res[k], self.msc_out[k],
                                   4 * self.n_sub * 192, self.msc_res[k], stream=self.streams[k].cuda_stream, bits_layout=self.layout)

    def step(self, on_demod=None):
        """one transmission frame of every ensemble: demod -> FIC + MSC"""
        torch, j, n = self.torch, self.j, self.inflight
        k, slot, st = j % n, j % self.H, self.streams[j % n]
        self.j += 1
        self.last_frame_of_lane = getattr(self, "last_frame_of_lane", {})
        self.last_frame_of_lane[k] = j
        if n == 1:
            if on_demod:
                on_demod(lambda: self.demod(slot, 0, j))
            else:
                self.demod(slot, 0, j)
            self.decode(slot)
            return
        w = self.ev_msc.pop(j - self.H + 4, None)                    # the last reader of the slot this frame overwrites
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
        self.decode(slot, k)
        ev = torch.cuda.Event(); ev.record(st)
        self.ev_msc[j] = ev

    def fill(self):
        for _ in range(self.H + self.inflight):      # fill the history ring: the time de-interleaver needs 16 CIFs = 4 frames
            self.step()
        self.torch.cuda.synchronize()

    def timed(self, fn, reps):
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            fn(k % self.H)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def check(self, dabgpu):
        """the outputs of the last frame of every lane against what was transmitted: frame j carries fibs[j mod 2], its CIF c decodes to
        payload[(4 j + c - 15) mod 8] -- the payload changes with every CIF, so a wrong ring slot, a wrong age or frames in flight in
        the wrong order cannot pass"""
        import numpy as np
        torch, E, nd, P = self.torch, self.E, self.mux.n, self.mux.period
        out = {"fib_crc_pass": 0, "fib_crc_expected": E * 12 * self.inflight, "fib_bytes_equal_transmitted": True,
               "msc_bytes_equal_transmitted": True, "ensembles_checked": E, "distinct_multiplexes": nd, "frames_in_flight": self.inflight,
               "payload_period_cifs": P, "frames_checked": []}
        idx = torch.arange(E, device=self.hist.device) % nd
        for k in range(self.inflight):                               # the outputs of the last frame of every lane
            j = self.last_frame_of_lane[k]
            out["frames_checked"].append(int(j))
            cifs = [(4 * j + c - 15) % P for c in range(4)]
            exp = self.mux.payload[idx][:, cifs]                      # [E, 4, n_sub, 192]
            res_f = self.fic_res[k].cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(E, 4)
            out["fib_crc_pass"] += int(np.unpackbits(res_f["crc_ok_mask"].astype("<u4").view(np.uint8)).sum())
            out["fib_bytes_equal_transmitted"] &= bool(torch.equal(self.fic_out[k], self.mux.fibs[idx, j % self.mux.n_frames]))
            out["msc_bytes_equal_transmitted"] &= bool(torch.equal(self.msc_out[k].view(E, 4, self.n_sub, 192), exp))
        return out


def extras_configs23(ctx, dabgpu, torch, device, E, n_distinct, reps=6, layout=1):
    """BASELINE configs[2] (demod + FIC Viterbi) and configs[3] (full FIC + MSC, E concurrent ensembles) on this GPU"""
    p = Pipeline(ctx, dabgpu, torch, device, E, n_distinct, seed=7, inflight=2, layout=layout)
    p.fill()
    torch.cuda.synchronize()
    t_demod, t_fic, t_msc = p.timed(p.demod, reps), p.timed(p.fic, reps), p.timed(p.msc, reps)      # one stage at a time, stream 0
    t_dec = p.timed(p.decode, reps)                                            # FIC + MSC as one call (what the pipeline runs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        p.demod(k % p.H); p.fic(k % p.H)
    e1.record(); torch.cuda.synchronize()
    t_c2_seq = e0.elapsed_time(e1) / reps
    # configs[2] with two frames in flight like configs[3]: frame j on lane j mod 2 (demod -> FIC in stream order; slot j mod 8 is reused
    # by frame j + 8 on the same lane), so the FIC trellis of frame j (two wavefronts per SIMD) runs beside the demodulation of j + 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4 * reps):
        p.demod(k % p.H, k % 2, k); p.fic(k % p.H, k % 2)
    torch.cuda.synchronize()
    t_c2 = (time.perf_counter() - t0) / (4 * reps) * 1e3
    t0 = time.perf_counter()
    for k in range(reps):                                     # one frame at a time on one stream
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
    # DABGPU_VIT_MAP_AUTO's switch points for the FIC (include/dabgpu.h): wave -> octet at ~700 frames, octet -> lane at ~12000
    lanes_fic = 64 if E >= 12000 else (8 if E >= 700 else 0)
    fic_kernel = {64: "vit_lanes_kernel (FIC)", 8: "vit_octet_kernel (FIC)", 0: "viterbi_kernel (FIC)"}[lanes_fic]
    c2 = {"workload": f"BASELINE configs[2]: full OFDM demod + FIC Viterbi (4 x 774 trellis steps per frame), {E} frames", "frames": E,
          "ms_per_step": t_c2, "frames_per_s": E / t_c2 * 1e3, "x_realtime": E / t_c2 * 1e3 / REALTIME_FRAMES_PER_S,
          "frames_in_flight": 2, "ms_per_step_one_frame_at_a_time": t_c2_seq, "frames_per_s_one_frame_at_a_time": E / t_c2_seq * 1e3,
          "kernel_ms": {"ofdm_demod": t_demod, "fic_viterbi": t_fic},
          "roofline": [hbm_roofline("ofdm_demod_kernel", t_demod, E),
                       viterbi_roofline(fic_kernel, p.fic_steps, t_fic, lanes_fic)],
          "check": {k: chk[k] for k in ("fib_crc_pass", "fib_crc_expected", "fib_bytes_equal_transmitted")}}
    c3 = {"workload": f"BASELINE configs[3]: full FIC + MSC demod + Viterbi, {E} concurrent synthetic ensembles, 18 x 48 CU EEP 3-A", "ensembles": E,
          "ms_per_step": t_all, "frames_per_s": E / t_all * 1e3, "x_realtime": E / t_all * 1e3 / REALTIME_FRAMES_PER_S,
          "frames_in_flight": 2, "history_layout": "time-interleaver class order" if layout else "natural",
          "ms_per_step_one_frame_at_a_time": t_seq, "frames_per_s_one_frame_at_a_time": E / t_seq * 1e3,
          "kernel_ms": {"ofdm_demod": t_demod, "fic_viterbi": t_fic, "msc_viterbi_incl_deinterleave": t_msc,
                        "fic_and_msc_one_call": t_dec},
          "algorithmic_hbm_GBps": 2.27e6 * E / (t_all * 1e-3) / 1e9,
          "roofline": [hbm_roofline("ofdm_demod_kernel", t_demod, E),
                       viterbi_roofline(("vit_prep_ring4c_kernel" if layout else "vit_prep_ring4_kernel") + " + vit_lanes_kernel (MSC)", p.msc_steps, t_msc, True)],
          "check": chk}
    del p
    torch.cuda.empty_cache()
    return c2, c3


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
    shard.barrier(dist if world > 1 else None)
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None)
    covered = shard.sum_over_ranks(n, dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "dab_mode1_frames_per_sec", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run (no kernels)",
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
        # symbols per workgroup: --spb 0 (default) leaves the choice to the library, which times a whole frame (75: one round of
        # workgroups on a full chip, phase tail inside the kernel), two and three runs per frame (38, 25) once per context and batch
        # size and keeps the fastest (include/dabgpu.h; which one wins depends on the box).  The timed loop below runs what
        # the library chose; the three are timed here once more (untimed region) only to put the numbers in the line
        spb_timing = None
        if args.spb == 0 and F >= 512 and not args.dry_run:
            ctx.ofdm_demod_phase_frames(iq_f, fmt_f32, F, d_bits, freq_offset=d_freq, cp_corr=d_corr, symbols_per_block=0,
                                        beta=0.9, total_phase=d_total, fine_freq=d_fine)          # first call: the library's calibration
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
                        layout=int(args.hist_layout == "classed"))
        pipe.fill()
        units = E

        def timed_demod(launch):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = pipe.streams[(pipe.j - 1) % pipe.inflight]
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
            chosen = args.spb or ctx.ofdm_auto_symbols_per_block(units) or 25
            config = {"workload": workload, "frames_per_gpu_per_step": units, "symbols_per_block": chosen,
                      "symbols_per_block_chosen_by": "caller (--spb)" if args.spb else "library (symbols_per_block = 0: one-time calibration per context and batch size)",
                      "sharding": "independent frames / ensembles per rank, no data-path collective"}
            if spb_timing:
                config["symbols_per_block_timed_ms"] = {str(k): v for k, v in spb_timing.items()}
        else:
            workload = (f"BASELINE configs[4]: {units * world} synthetic ensembles ({units} per GPU, built on the device from "
                        f"{min(args.distinct, units)} seeded multiplexes), full FIC+MSC: OFDM demod + FIC Viterbi + 18 x 48 CU EEP 3-A "
                        "time de-interleave / Viterbi / descramble per transmission frame")
            config = {"workload": workload, "ensembles_per_gpu": units, "frames_per_gpu_per_step": units,
                      "sharding": "independent ensembles per rank, no data-path collective"}
        line = {
            "metric": "dab_mode1_frames_per_sec", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config,
            "x_realtime": value / REALTIME_FRAMES_PER_S,
            "roofline": hbm_roofline("ofdm_demod_kernel", k_ms, units),
            "check": check,
        }
        if region:
            line["roofline"]["timing"] = (f"one pair of HIP events around the {args.steps} back-to-back launches of the timed loop, on their stream: "
                                          f"{k_ms_events:.4f} ms per launch" + ("" if (args.spb or ctx.ofdm_auto_symbols_per_block(units)) == 75 else " (incl. the 5 us phase-tail launch of each step)"))
        else:
            # configs[4]: inside the timed loop the demodulation of frame j + 1 runs BESIDE the trellis kernel of frame j (two frames in
            # flight), so its launches last longer than the kernel needs; the rooflines are taken one stage at a time after the loop
            pipe.timed(pipe.msc, 4); pipe.timed(pipe.demod, 8)              # (the clock has dropped during the host-side check above)
            t_d, t_f, t_m = pipe.timed(pipe.demod, 6), pipe.timed(pipe.fic, 6), pipe.timed(pipe.msc, 6)
            line["roofline"] = hbm_roofline("ofdm_demod_kernel", t_d, units)
            line["roofline"]["timing"] = (f"one stage at a time after the timed loop: HIP events around 6 launches, {t_d:.4f} ms each (inside the loop, "
                                          f"overlapped with the other frame's trellis kernel: {k_ms_events:.4f} ms, mean of {len(evs)} launches)")
            line["roofline_decode"] = viterbi_roofline(("vit_prep_ring4c_kernel" if pipe.layout else "vit_prep_ring4_kernel") + " + vit_lanes_kernel (MSC)",
                                                       pipe.msc_steps, t_m, True)
            line["stage_ms_one_at_a_time"] = {"ofdm_demod": t_d, "fic_viterbi": t_f, "msc_viterbi_incl_deinterleave": t_m}
        # PMC-derived HBM traffic per launch, when a profiles/ summary of this round exists (see profiles/README.md)
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
                tr = json.load(fh)
            if tr.get("frames_per_launch") == units:
                line["roofline"]["traffic"] = tr.get("bytes_per_launch")
                line["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass
        if world == 1 and args.workload == "demod" and not args.no_extras:
            del iq, iq_f, tx_bits, d_bits
            torch.cuda.empty_cache()
            c2, c3 = extras_configs23(ctx, dabgpu, torch, device, args.extra_ensembles, args.distinct, layout=int(args.hist_layout == "classed"))
            line["extra"] = {"configs2": c2, "configs3": c3}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
