#!/usr/bin/env python3
"""Measures BASELINE configs[2] and configs[3] (not the driver's bench contract -- that is bench.py):
full OFDM demod + FIC Viterbi, and FIC + MSC for E concurrent synthetic ensembles with the canonical multiplex of
SURVEY 8(d) (18 sub-channels x 48 CU, EEP 3-A, all 864 CU), on one MI355X.

Every ensemble carries its own CRC-valid FIBs and sub-channel payloads, channel-coded with torch on the device
(generator side, untimed); every CIF of an ensemble repeats the same logical frames so that the 16-CIF time
interleaver is in steady state from one stored frame.  Self-check: FIB CRC pass count and decoded bytes == payload.

    python tools/bench_decode.py [--ensembles 1024] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dabgpu  # noqa: E402

TAPS = [[0, 2, 3, 5, 6], [0, 1, 2, 3, 6], [0, 1, 4, 6], [0, 2, 3, 5, 6]]        # polynomials 133,171,145,133 (octal)


def prbs_bytes(n):
    reg, out = 0xFFFF, np.empty(n, np.uint8)
    for k in range(n):
        b = 0
        for i in range(8):
            v = ((reg >> 8) ^ (reg >> 4)) & 1
            b |= v << (7 - i)
            reg = ((reg << 1) | v) & 0xFFFF
        out[k] = b
    return out


def kept_index(segments):
    """mother-code bit indices that survive puncturing for [(PI, L blocks)...] + the PI_X tail"""
    order = [0, 4, 2, 6, 1, 5, 3, 7]
    idx, m = [], 0
    for pi, L in segments + [(8, None)]:
        cnt = [1] * 8
        for e in range(pi):
            cnt[order[e % 8]] += 1
        n_groups = 6 if L is None else 32 * L
        for g in range(n_groups):
            idx += [m + 4 * g + r for r in range(cnt[g % 8])]
        m += 4 * n_groups
    return np.array(idx, dtype=np.int64)


def bytes_to_bits(x):
    sh = torch.arange(7, -1, -1, device=x.device, dtype=torch.uint8)
    return ((x.unsqueeze(-1) >> sh) & 1).reshape(*x.shape[:-1], -1)


def conv_encode(bits):
    """bits [..., n] (0/1 uint8) -> mother code [..., 4*(n+6)]"""
    n = bits.shape[-1]
    x = torch.nn.functional.pad(bits, (6, 6))
    outs = []
    for taps in TAPS:
        acc = torch.zeros(bits.shape[:-1] + (n + 6,), dtype=torch.uint8, device=bits.device)
        for d in taps:
            acc ^= x[..., 6 - d: 6 - d + n + 6]
        outs.append(acc)
    return torch.stack(outs, dim=-1).reshape(*bits.shape[:-1], -1)


def crc16(data):
    """data [..., nbytes] uint8 -> [..., 2] bytes (poly 0x1021, init/xorout 0xFFFF)"""
    bits = bytes_to_bits(data).to(torch.int32)
    crc = torch.full(data.shape[:-1], 0xFFFF, dtype=torch.int32, device=data.device)
    for i in range(bits.shape[-1]):
        msb = ((crc >> 15) & 1) ^ bits[..., i]
        crc = ((crc << 1) & 0xFFFF) ^ (msb * 0x1021)
    crc ^= 0xFFFF
    return torch.stack([(crc >> 8) & 0xFF, crc & 0xFF], dim=-1).to(torch.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ensembles", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--tie-rule", type=int, default=0)
    ap.add_argument("--hist-layout", choices=("classed", "natural"), default="classed",
                    help="order of the MSC soft bits in the frame-history ring (classed = DABGPU_BITS_MSC_CLASSED)")
    ap.add_argument("--mapping", type=int, default=0, help="0 auto, 1 wave per codeword, 2 lane per codeword, 3 eight lanes per codeword (DABGPU_VIT_MAP_*)")
    ap.add_argument("--spb", type=int, default=0, help="data symbols per demodulator workgroup (0 = the library's own choice; counter passes fix it so that every launch is alike)")
    ap.add_argument("--no-overlap", action="store_true", help="serial stages only (profiling passes: every launch runs alone)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = dabgpu.Context(0)
    ctx.viterbi_set_mapping(args.mapping)
    prs, mapper, _ = dabgpu.host_tables()
    E, H = args.ensembles, 5
    g = torch.Generator(device=dev)
    g.manual_seed(7)

    # ---- transmit side (untimed): FIC ----
    fib_data = torch.randint(0, 256, (E, 4, 3, 30), generator=g, device=dev, dtype=torch.uint8)
    fibs = torch.cat([fib_data, crc16(fib_data)], dim=-1).reshape(E, 4, 96)
    pr96 = torch.from_numpy(prbs_bytes(96)).to(dev)
    fic_mother = conv_encode(bytes_to_bits(fibs ^ pr96))
    fic_tx = fic_mother[..., torch.from_numpy(kept_index([(16, 21), (15, 3)])).to(dev)]            # [E,4,2304]
    # ---- MSC: 18 x 48 CU EEP 3-A: PI_8 x 45 blocks, PI_7 x 3 blocks, 192 bytes per CIF ----
    n_sub = 18
    payload = torch.randint(0, 256, (E, n_sub, 192), generator=g, device=dev, dtype=torch.uint8)
    pr192 = torch.from_numpy(prbs_bytes(192)).to(dev)
    kidx = torch.from_numpy(kept_index([(8, 45), (7, 3)])).to(dev)
    msc_tx = torch.empty((E, n_sub, 3072), dtype=torch.uint8, device=dev)
    for e0 in range(0, E, 256):
        msc_tx[e0:e0 + 256] = conv_encode(bytes_to_bits(payload[e0:e0 + 256] ^ pr192))[..., kidx]
    cif = msc_tx.reshape(E, 55296)
    frame_bits = torch.cat([fic_tx.reshape(E, 9216), cif.repeat(1, 4)], dim=1).reshape(E, 75, 3072)

    # ---- OFDM modulation of one frame per ensemble (as bench.py, without CFO to keep generation light) ----
    mp = torch.from_numpy(mapper.astype(np.int64)).to(dev)
    bins = torch.where(mp < 768, mp + (2048 - 768), mp - 768 + 1)
    prs_t = torch.from_numpy(prs).to(dev)
    iq = torch.zeros((E, 196608), dtype=torch.complex64, device=dev)
    a = 0.70710678
    for e0 in range(0, E, 32):
        b = frame_bits[e0:e0 + 32]
        m = b.shape[0]
        z = torch.complex((1.0 - 2.0 * b[:, :, :1536].float()) * a, (1.0 - 2.0 * b[:, :, 1536:].float()) * a)
        spec = torch.zeros((m, 76, 2048), dtype=torch.complex64, device=dev)
        spec[:, 0] = prs_t
        cur = prs_t[bins].expand(m, -1).clone()
        for s_ in range(75):
            cur = cur * z[:, s_]
            spec[:, s_ + 1, bins] = cur
        t = torch.fft.ifft(spec, dim=2) * (2048.0 / 39.2)
        body = iq[e0:e0 + m, : 76 * 2552].view(m, 76, 2552)
        body[:, :, 504:] = t
        body[:, :, :504] = t[:, :, 2048 - 504:]
    iq += 0.05 * torch.randn(iq.shape, dtype=torch.complex64, device=dev)
    iq_f = torch.view_as_real(iq)

    hist = torch.zeros((E, H, 230400), dtype=torch.int8, device=dev)
    LAYOUT, F32 = int(args.hist_layout == "classed"), dabgpu.IQ_FORMATS.index("raw_f32l")
    corr = torch.empty((E, 76, 2), dtype=torch.float32, device=dev)
    fic_out = torch.zeros((E, 4, 96), dtype=torch.uint8, device=dev)
    fic_res = torch.zeros((E * 4, 16), dtype=torch.uint8, device=dev)
    msc_out = torch.zeros((E, 4, n_sub * 192), dtype=torch.uint8, device=dev)
    msc_res = torch.zeros((E * 4 * n_sub, 16), dtype=torch.uint8, device=dev)
    subs = [dabgpu.SubChannel(48 * s, 48, 0, 0, 2, 0) for s in range(n_sub)]

    def demod(slot):
        ctx.ofdm_demod_frames_history(iq_f, F32, E, hist[:, slot], cp_corr=corr, symbols_per_block=args.spb, bits_frame_stride=H * 230400, bits_layout=LAYOUT)

    def fic(slot):
        ctx.fic_decode_frames(hist[:, slot], E, fic_out, fic_res, frame_stride=H * 230400, tie_rule=args.tie_rule)

    def msc(slot):
        ctx.msc_decode_frames(hist, E, H * 230400, H, slot, subs, msc_out, 4 * n_sub * 192, msc_res, tie_rule=args.tie_rule, bits_layout=LAYOUT)

    for slot in range(H):                   # fill the history ring (and warm up)
        demod(slot)
    fic(0); msc(0)
    torch.cuda.synchronize()

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            fn(k % H)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_demod = timed(demod, args.steps)
    t_fic = timed(fic, args.steps)
    t_msc = timed(msc, args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        demod(k % H); fic(k % H); msc(k % H)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / args.steps * 1e3

    if args.no_overlap:
        print(json.dumps({"ensembles": E, "kernel_ms": {"ofdm_demod": t_demod, "fic_viterbi": t_fic, "msc_viterbi": t_msc}, "ms_per_frame_step_wall": t_all}))
        return

    # the same with the FIC on a second stream: it decodes concurrently with the MSC of the same frames (separate scratch,
    # include/dabgpu.h) and fits into the SIMD time the MSC's long trellis waves leave idle
    s_fic = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def step_two_streams(slot):
        demod(slot)
        ev = torch.cuda.Event()
        ev.record(main)
        s_fic.wait_event(ev)
        ctx.fic_decode_frames(hist[:, slot], E, fic_out, fic_res, frame_stride=H * 230400, tie_rule=args.tie_rule, stream=s_fic.cuda_stream)
        msc(slot)
        ev2 = torch.cuda.Event()
        ev2.record(s_fic)
        main.wait_event(ev2)

    step_two_streams(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_two_streams(k % H)
    torch.cuda.synchronize()
    t_all2 = (time.perf_counter() - t0) / args.steps * 1e3

    # frames in flight on two streams (two contexts: each owns its scratch): even frames on one, odd frames on the other, history ring of
    # H2 = 8 slots so that the demodulator of frame j + 1 never writes a slot the MSC decoder of frame j still reads (it reads
    # frames j - 4 .. j).  The next frame's HBM-bound demod / gather kernels fill the wavefront slots the trellis kernel's last,
    # partial round leaves idle.  Dependencies: msc(j) needs the ring slots written by demod(j - 4 .. j) -> wait for demod(j - 1).
    H2 = 8
    hist2 = torch.zeros((E, H2, 230400), dtype=torch.int8, device=dev)
    ctxs = [ctx, dabgpu.Context(0)]
    ctxs[1].viterbi_set_mapping(args.mapping)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    fic_out2 = [fic_out, torch.zeros_like(fic_out)]
    fic_res2 = [fic_res, torch.zeros_like(fic_res)]
    msc_out2 = [msc_out, torch.zeros_like(msc_out)]
    msc_res2 = [msc_res, torch.zeros_like(msc_res)]
    corr2 = [corr, torch.empty_like(corr)]

    def frame_on(j, ev_prev_demod):
        k = j & 1
        c, st = ctxs[k], streams[k]
        slot = j % H2
        with torch.cuda.stream(st):
            c.ofdm_demod_frames_history(iq_f, F32, E, hist2[:, slot], cp_corr=corr2[k], symbols_per_block=args.spb, bits_frame_stride=H2 * 230400, bits_layout=LAYOUT,
                                        stream=st.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(st)
            c.fic_decode_frames(hist2[:, slot], E, fic_out2[k], fic_res2[k], frame_stride=H2 * 230400, tie_rule=args.tie_rule, stream=st.cuda_stream)
            if ev_prev_demod is not None:
                st.wait_event(ev_prev_demod)
            c.msc_decode_frames(hist2, E, H2 * 230400, H2, slot, subs, msc_out2[k], 4 * n_sub * 192, msc_res2[k], tie_rule=args.tie_rule,
                                stream=st.cuda_stream, bits_layout=LAYOUT)
        return ev

    torch.cuda.synchronize()
    ev = None
    for j in range(H2):                       # fill the longer ring
        ev = frame_on(j, ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = None
    n_pp = 2 * args.steps
    for j in range(n_pp):
        ev = frame_on(H2 + j, ev)
    torch.cuda.synchronize()
    t_pp = (time.perf_counter() - t0) / n_pp * 1e3
    pp_ok = bool(torch.equal(msc_out2[1].view(E, 4, n_sub, 192), payload.unsqueeze(1).expand(E, 4, n_sub, 192))) and bool(torch.equal(fic_out2[1], fibs))

    res_f = fic_res.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(E, 4)
    crc_ok = int(sum(bin(int(m)).count("1") for m in res_f["crc_ok_mask"].reshape(-1)))
    fib_eq = bool(torch.equal(fic_out, fibs))
    msc_eq = bool(torch.equal(msc_out.view(E, 4, n_sub, 192), payload.unsqueeze(1).expand(E, 4, n_sub, 192)))
    fic_steps, msc_steps = 4 * 774, 4 * n_sub * 1542
    out = {
        "ensembles": E, "multiplex": "18 x 48 CU EEP 3-A (864 CU)", "tie_rule": args.tie_rule, "viterbi_mapping": args.mapping,
        "config3_demod_plus_fic": {"ms_per_frame_step": t_demod + t_fic, "frames_per_s": E / (t_demod + t_fic) * 1e3,
                                   "fic_trellis_steps_per_s": E * fic_steps / t_fic * 1e3},
        "config4_full": {"ms_per_frame_step_sum_of_kernels": t_demod + t_fic + t_msc, "ms_per_frame_step_wall": t_all,
                         "frames_per_s": E / t_all * 1e3, "x_realtime": E / t_all * 1e3 / (2.048e6 / 196608),
                         "ms_per_frame_step_wall_fic_on_second_stream": t_all2, "frames_per_s_fic_on_second_stream": E / t_all2 * 1e3,
                         "ms_per_frame_step_two_frames_in_flight": t_pp, "frames_per_s_two_frames_in_flight": E / t_pp * 1e3, "two_frames_in_flight_outputs_ok": pp_ok,
                         "msc_trellis_steps_per_s": E * msc_steps / t_msc * 1e3},
        "kernel_ms": {"ofdm_demod": t_demod, "fic_viterbi": t_fic, "msc_viterbi": t_msc},
        "history_layout": args.hist_layout, "symbols_per_block_chosen_by_library": ctx.ofdm_auto_symbols_per_block(E),
        "check": {"fib_crc_pass": crc_ok, "fib_crc_expected": E * 12, "fib_bytes_equal_transmitted": fib_eq,
                  "msc_bytes_equal_transmitted": msc_eq},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
