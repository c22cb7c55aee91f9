#!/usr/bin/env python3
"""Throughput of the DAB+ outer-code kernel (dabgpu_dabplus_bank_process) on one MI355X: S streams (ensemble x sub-channel
pairs), each fed 5 logical frames = one audio super frame per call, with a chosen number of symbol errors in every
RS(120,110) codeword.  Self-checking: every super frame must come out with a valid header and all access-unit CRCs.

    python tools/bench_dabplus.py [--streams 18432] [--frame-bytes 192] [--reps 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dabgpu  # noqa: E402

# GF(2^8), p(x) = x^8+x^4+x^3+x^2+1 (ETSI TS 102 563 clause 6.1) -- generator side only
EXP = np.zeros(512, np.int64)
LOG = np.zeros(256, np.int64)
_x = 1
for _i in range(255):
    EXP[_i] = EXP[_i + 255] = _x
    LOG[_x] = _i
    _x <<= 1
    if _x & 0x100:
        _x ^= 0x11D


def gmul(a, b):
    return 0 if a == 0 or b == 0 else int(EXP[LOG[a] + LOG[b]])


def rs_parity(data):
    g = [1] + [0] * 10
    for i in range(10):
        for j in range(i + 1, 0, -1):
            g[j] = g[j - 1] ^ gmul(g[j], int(EXP[i]))
        g[0] = gmul(g[0], int(EXP[i]))
    rem = [0] * 10
    for d in data:
        fb = int(d) ^ rem[9]
        for i in range(9, 0, -1):
            rem[i] = rem[i - 1] ^ gmul(fb, g[i])
        rem[0] = gmul(fb, g[0])
    return rem[::-1]


def crc16(data, poly, init, xorout):
    crc = init
    for b in data:
        crc ^= int(b) << 8
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ xorout


def make_superframe(rng, n):
    n_rs = 5 * n // 120
    data_len = 110 * n_rs
    sf = np.zeros(5 * n, np.uint8)
    sf[2] = 0x60                                         # 48 kHz, SBR: 3 access units
    first = 3 + 3
    cuts = [first, first + (data_len - first) // 3, first + 2 * (data_len - first) // 3, data_len]
    bits = []
    for v in cuts[1:3]:
        bits += [(v >> (11 - b)) & 1 for b in range(12)]
    sf[3:6] = np.packbits(np.array(bits, np.uint8))
    for i in range(3):
        a, b = cuts[i], cuts[i + 1]
        sf[a:b - 2] = rng.integers(0, 256, b - a - 2, dtype=np.uint8)
        c = crc16(sf[a:b - 2], 0x1021, 0xFFFF, 0xFFFF)
        sf[b - 2], sf[b - 1] = c >> 8, c & 0xFF
    fc = crc16(sf[2:11], 0x782F, 0, 0)
    sf[0], sf[1] = fc >> 8, fc & 0xFF
    for i in range(n_rs):
        sf[i + 110 * n_rs::n_rs] = rs_parity(sf[i:i + 110 * n_rs:n_rs])
    return sf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=18432)
    ap.add_argument("--frame-bytes", type=int, default=192)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    n, S = args.frame_bytes, args.streams
    n_rs = 5 * n // 120
    rng = np.random.default_rng(1)
    base = np.stack([make_superframe(rng, n) for _ in range(32)])
    ctx = dabgpu.Context(0)
    out = {"streams": S, "frame_bytes": n, "rs_codewords_per_superframe": n_rs, "cases": {}}
    for errs in (0, 2, 5):
        sfs = base[rng.integers(0, 32, S)].copy()
        for k in range(errs):                            # `errs` distinct symbols of every codeword
            rows = np.arange(S)[:, None]
            # (not in the first bytes of the super frame: a damaged fire code would only delay acquisition to the next call)
            col = (rng.integers(2, 24, (S, n_rs)) + 24 * k) * n_rs + np.arange(n_rs)[None, :]
            sfs[rows, col] ^= rng.integers(1, 256, (S, n_rs), dtype=np.uint8)
        d_frames = torch.from_numpy(sfs).cuda()          # [S][5][n]
        d_off = (torch.arange(S, dtype=torch.int64, device="cuda") * (5 * n))
        d_n = torch.full((S,), n, dtype=torch.int32, device="cuda")
        d_sf = torch.zeros((S, 1, 5 * n), dtype=torch.uint8, device="cuda")
        rec_bytes = np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE).itemsize
        d_res = torch.zeros((S, 1, rec_bytes), dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros((S, 4), dtype=torch.int32, device="cuda")
        bank = dabgpu.DabPlusBank(ctx, S)

        def run():
            bank.process(d_frames, d_off, n, d_n, 5, d_sf, 5 * n, d_res, 1, d_cnt)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        res = d_res.cpu().numpy().view(np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)).reshape(S)
        ok = bool((res["header_valid"] == 1).all() and (res["au_crc_ok_mask"] == 7).all() and (res["rs_corrected"] == errs * n_rs).all()
                  and (d_cnt[:, 0] == 1).all().item())
        out["cases"][f"{errs}_errors_per_codeword"] = {"ms": ms, "superframes_per_s": S / ms * 1e3, "rs_codewords_per_s": S * n_rs / ms * 1e3,
                                                      "x_realtime_with_all_streams_concurrent": 120.0 / ms, "all_valid": ok}
        bank.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
