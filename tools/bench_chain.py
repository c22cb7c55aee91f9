#!/usr/bin/env python3
"""The device-resident chain at full size: E unsynchronised raw_u8 capture streams (each receiver starts somewhere inside the
transmission, with its own carrier offset and noise) ->
    dabgpu_stream_bank_process_ring_layout   NULL search, PRS synchronisation, framing, demodulation into per-ensemble history rings
 -> dabgpu_decode_ring_layout                FIC + time de-interleave + Viterbi + descrambling of 18 sub-channels, each ensemble at its own ring slot
 -> dabgpu_dabplus_bank_process_masked       fire code, super-frame collection, RS(120,110), access-unit CRCs of the 18 DAB+ streams
with nothing crossing PCIe between capture bytes in and access units out -- the chain tests/test_gpu_device_pipeline.py proves against
the oracle on three ensembles, here timed on 4096 (BASELINE configs[3]'s ensemble count) and checked against what was transmitted.

One step = one block of 191,400 samples per receiver (the most a ring call takes: at most one frame completes per stream); the rate is
frames completed per second over the timed steps (read from the bank's counters before and after, untimed).
`overlap`: the bank runs on one HIP stream, the decoders + DAB+ on another (own context), so block j + 1's front end runs beside block
j's trellis kernel; `sequential`: everything on one stream.

    python tools/bench_chain.py [--ensembles 4096] [--steps 12] [--distinct 64]
(bench.py calls run_chain() for its `extra.chain` block.)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

BLOCK = 191400                       # include/dabgpu.h: at most 191400 samples per dabgpu_stream_bank_process_ring call
REALTIME_FRAMES_PER_S = 2.048e6 / 196608


def run_chain(ctx, dabgpu, torch, device, E, n_distinct, steps=12, warm_steps=8, seed=31, layout=1, rs_errors=1, with_dabplus=True, noise=0.05,
              sequential_steps=4, retained=True):
    import numpy as np
    import dabsynth
    prs, mapper, _ = dabgpu.host_tables()
    H, nsub, nb = 8, dabsynth.N_SUB, dabsynth.SUB_BYTES
    t_gen = time.perf_counter()
    # 20 CIFs = 5 transmission frames repeat: whole DAB+ super frames (5 logical frames) and a legal time-interleaver steady state
    iq, mux = dabsynth.ensemble_iq(E, min(n_distinct, E), seed, device, mapper, prs, noise=0.0, period=20, superframes=True, rs_errors=rs_errors)
    nf = mux.n_frames
    rep = nf * dabsynth.NB_FRAME_SAMPLES
    raw, start, cfo = dabsynth.ensemble_streams_u8(iq, seed + 1, BLOCK, noise=noise)
    del iq
    torch.cuda.empty_cache()
    t_gen = time.perf_counter() - t_gen
    stride = raw.shape[1]                                        # samples per receiver buffer (rep + BLOCK)
    fmt_u8 = dabgpu.IQ_FORMATS.index("raw_u8")
    bank = dabgpu.StreamBank(ctx, E)
    ctx2 = dabgpu.Context(device.index)
    dp = dabgpu.DabPlusBank(ctx2, E * nsub) if with_dabplus else None
    subs = mux.subchannels(dabgpu)
    hist = torch.zeros((E, H, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=device)
    slots = [torch.full((E,), -1, dtype=torch.int32, device=device) for _ in range(2)]
    fib = torch.zeros((E, 4, 96), dtype=torch.uint8, device=device)
    fres = torch.zeros((E * 4, 16), dtype=torch.uint8, device=device)
    msc = torch.zeros((E, 4, nsub * nb), dtype=torch.uint8, device=device)
    mres = torch.zeros((E * 4 * nsub, 16), dtype=torch.uint8, device=device)
    rec_dt = np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)
    S = E * nsub                                                 # DAB+ streams, ensemble-major
    sf = torch.zeros((S, 1, 5 * nb), dtype=torch.uint8, device=device)
    rec = torch.zeros((S, 1, rec_dt.itemsize), dtype=torch.uint8, device=device)
    cnt = torch.zeros((S, 4), dtype=torch.int32, device=device)
    e_idx = torch.arange(E, dtype=torch.int64, device=device)
    d_off = (e_idx[:, None] * (4 * nsub * nb) + torch.arange(nsub, dtype=torch.int64, device=device)[None, :] * nb).reshape(-1).contiguous()
    d_nb = torch.full((S,), nb, dtype=torch.int32, device=device)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    pos = [0]
    prev = [None]
    ev_dec = [None, None]

    def front(j, stream):
        """block j of every receiver through the stream bank (synchronises with `stream`: the number of rounds depends on the data).
        retained: the capture buffer outlives every call (a ring of device buffers does, dabgpu_ingest_*), so the bank leaves the
        unfinished frame at a block's end where it is (dabgpu_stream_bank_process_ring_retained)"""
        p = pos[0]
        pos[0] = (p + BLOCK) % rep
        if retained:
            bank.process_ring_retained(raw.data_ptr() + 2 * p, fmt_u8, stride, BLOCK, prev[0], hist, H, slots[j % 2], stream=stream.cuda_stream, bits_layout=layout)
            prev[0] = raw.data_ptr() + 2 * p
        else:
            bank.process_ring(raw.data_ptr() + 2 * p, fmt_u8, stride, BLOCK, hist, H, slots[j % 2], stream=stream.cuda_stream, bits_layout=layout)

    def back(j, stream):
        c = ctx2
        c.decode_ring(hist, E, H * dabgpu.NB_FRAME_BITS, H, slots[j % 2], subs, fib, fres, msc, 4 * nsub * nb, mres, stream=stream.cuda_stream, bits_layout=layout)
        if dp is not None:
            dp.process_masked(msc, d_off, nsub * nb, d_nb, 4, sf, 5 * nb, rec, 1, cnt, slots[j % 2], nsub, stream=stream.cuda_stream)

    def step_overlapped(j):
        if ev_dec[j % 2] is not None:                            # the decoders of block j - 2 have read this slot array
            sA.wait_event(ev_dec[j % 2])
        front(j, sA)
        e = torch.cuda.Event(); e.record(sA)
        sB.wait_event(e)
        back(j, sB)
        e2 = torch.cuda.Event(); e2.record(sB)
        ev_dec[j % 2] = e2

    def step_sequential(j):
        front(j, sA)
        back(j, sA)

    def frames_read():
        return int(bank.status()["total_frames_read"].astype(np.int64).sum())

    j = 0
    for _ in range(warm_steps):                                  # acquisition (NULL search, first sync), time de-interleaver and super-frame sync fill
        step_overlapped(j); j += 1
    torch.cuda.synchronize()
    f0 = frames_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_overlapped(j); j += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f1 = frames_read()
    # the same on one stream
    t0 = time.perf_counter()
    for _ in range(sequential_steps):
        step_sequential(j); j += 1
    torch.cuda.synchronize()
    dt_seq = time.perf_counter() - t0
    f2 = frames_read()
    # ---- check (untimed): more steps, read back after each: every frame completed in them must decode to what was transmitted ----
    status0 = bank.status()
    chk = {"frames_checked": 0, "fib_crc_pass": 0, "fib_crc_expected": 0, "fib_bytes_equal_transmitted": True, "msc_bytes_equal_transmitted": True,
           "superframes_checked": 0, "superframes_all_valid": True, "rs_symbols_corrected": 0, "superframe_bytes_equal_transmitted": True,
           "streams_locked": int((status0["state"] == 4).sum() + (status0["state"] == 1).sum()), "streams": E,
           "desyncs_total": int(status0["total_frames_desync"].astype(np.int64).sum())}
    idx = (e_idx % mux.n)
    clean = torch.from_numpy(mux.superframes_clean).to(device)                                     # [32][960]
    pick = torch.from_numpy(mux.superframe_pick).to(device)                                        # [n][sub][period / 5]
    for _ in range(6):
        cnt.zero_()                                                  # (the bank writes a stream's counters only when it processes the stream)
        step_sequential(j); j += 1
        torch.cuda.synchronize()
        sl = slots[(j - 1) % 2]
        done = sl >= 0
        nd = int(done.sum().item())
        if nd == 0:
            continue
        chk["frames_checked"] += nd
        res_f = fres.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(E, 4)
        dm = done.cpu().numpy()
        chk["fib_crc_pass"] += int(np.unpackbits(res_f["crc_ok_mask"][dm].astype("<u4").view(np.uint8)).sum())
        chk["fib_crc_expected"] += 12 * nd
        # which of the nf stored frames did each receiver just complete?  the one whose FIBs it decoded
        eq = (fib[:, None] == mux.fibs[idx]).flatten(2).all(dim=2)                                  # [E][nf]
        chk["fib_bytes_equal_transmitted"] &= bool(eq.any(dim=1)[done].all().item())
        fsel = eq.float().argmax(dim=1)                                                             # [E]
        cifs = (4 * fsel[:, None] + torch.arange(4, device=device)[None, :] - 15) % mux.period      # [E][4]
        exp = mux.payload[idx[:, None], cifs]                                                       # [E][4][nsub][nb]
        chk["msc_bytes_equal_transmitted"] &= bool((msc.view(E, 4, nsub, nb) == exp).flatten(1).all(dim=1)[done].all().item())
        if dp is not None:
            c = cnt.cpu().numpy()
            r = rec.cpu().numpy().view(rec_dt).reshape(S)
            att = (c[:, 0] > 0) & np.repeat(dm, nsub)
            chk["superframes_checked"] += int(att.sum())
            if att.any():
                ok = (r["header_valid"][att] == 1) & (r["au_crc_ok_mask"][att] == 7) & (r["rs_failed_index"][att] == -1) & (r["firecode_ok"][att] == 1)
                chk["superframes_all_valid"] &= bool(ok.all())
                chk["rs_symbols_corrected"] += int(r["rs_corrected"][att].sum())
                # the corrected super frame is one of the generated ones: the one the multiplex put at that position
                att_t = torch.from_numpy(att).to(device)
                q = ((cifs[:, 0:1] + torch.from_numpy(r["frame_index"].reshape(E, nsub).astype(np.int64)).to(device)) % mux.period) // 5      # [E][nsub]
                want = clean[pick[idx[:, None], torch.arange(nsub, device=device)[None, :], q]]     # [E][nsub][960]
                same = (sf.view(E, nsub, 5 * nb) == want).all(dim=2).reshape(-1)
                chk["superframe_bytes_equal_transmitted"] &= bool(same[att_t].all().item())
    # stage times, one at a time -- AFTER the check: a front end run without its decoders drops logical frames the DAB+ stage is collecting
    def timed(fn, reps):
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0_) / reps * 1e3
    jj = [j]
    def only_front():
        front(jj[0], sA); jj[0] += 1
    t_front = timed(only_front, 4)
    j = jj[0]
    t_dec = timed(lambda: ctx2.decode_ring(hist, E, H * dabgpu.NB_FRAME_BITS, H, slots[(j - 1) % 2], subs, fib, fres, msc, 4 * nsub * nb, mres,
                                           stream=sA.cuda_stream, bits_layout=layout), 4)

    frames_timed = f1 - f0
    out = {"workload": f"device-resident chain, {E} unsynchronised raw_u8 streams ({mux.n} seeded multiplexes of 18 x 48 CU EEP 3-A DAB+ sub-channels, "
                       f"{rs_errors} damaged symbol(s) per RS codeword): stream bank -> history rings -> FIC + MSC ring decode -> DAB+ outer code",
           "ensembles": E, "block_samples": BLOCK, "steps": steps, "frames_completed_in_timed_steps": frames_timed,
           "ms_per_step": dt / steps * 1e3, "frames_per_s": frames_timed / dt, "x_realtime": frames_timed / dt / REALTIME_FRAMES_PER_S,
           "streams": "front end on one HIP stream, decoders + DAB+ on another (context of their own)",
           "frames_per_s_one_stream": (f2 - f1) / dt_seq, "ms_per_step_one_stream": dt_seq / sequential_steps * 1e3,
           "stage_ms_one_at_a_time": {"stream_bank_process_ring": t_front, "decode_ring_fic_and_msc": t_dec},
           "history_layout": "time-interleaver class order" if layout else "natural",
           "blocks": "retained (dabgpu_stream_bank_process_ring_retained: no carry-over copy)" if retained else "copied (dabgpu_stream_bank_process_ring_layout)",
           "input_bytes_per_frame": 2 * dabsynth.NB_FRAME_SAMPLES, "generation_s_untimed": t_gen, "check": chk}
    bank.close()
    if dp is not None:
        dp.close()
    ctx2.close()
    del raw, hist
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ensembles", type=int, default=4096)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--hist-layout", choices=("classed", "natural"), default="classed")
    ap.add_argument("--rs-errors", type=int, default=1)
    ap.add_argument("--no-dabplus", action="store_true")
    ap.add_argument("--copying", action="store_true", help="the ring form that copies the unfinished frame at a block's end (blocks need not outlive the call)")
    a = ap.parse_args()
    import torch
    import dabgpu
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = dabgpu.Context(0)
    out = run_chain(ctx, dabgpu, torch, device, a.ensembles, a.distinct, steps=a.steps, layout=int(a.hist_layout == "classed"),
                    rs_errors=a.rs_errors, with_dabplus=not a.no_dabplus, retained=not a.copying)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
