"""Generates tests/golden/reference_vectors.npz from objects compiled out of the reference's own
sources (oracle/_ref/libdab_ref.so, built by `make -C oracle ref` where /root/reference exists).

The fixture holds DATA only: seeded inputs and the outputs the reference code produced for them.
Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def main():
    R = O.ref()
    assert R is not None, "oracle/_ref/libdab_ref.so missing: needs /root/reference"
    out = {}
    # constant tables
    m = np.zeros(1536, np.int32); R.ref_get_mapper(m.ctypes.data, 1536, 2048); out["mapper"] = m.astype(np.int16)
    p = np.zeros(2048, np.complex64); R.ref_get_prs(1, p.ctypes.data, 2048); out["prs_fft"] = p
    op = np.zeros(6, np.uint64); R.ref_get_ofdm_params(1, op.ctypes.data); out["ofdm_params_mode1"] = op
    dp = np.zeros(13, np.int32); R.ref_get_dab_params(1, dp.ctypes.data); out["dab_params_mode1"] = dp
    for mode in (2, 3, 4):                                   # transmission modes II-IV: geometry and frequency interleaver
        opm = np.zeros(6, np.uint64); R.ref_get_ofdm_params(mode, opm.ctypes.data); out[f"ofdm_params_mode{mode}"] = opm
        mm = np.zeros(int(opm[5]), np.int32); R.ref_get_mapper(mm.ctypes.data, int(opm[5]), int(opm[4])); out[f"mapper_mode{mode}"] = mm.astype(np.int16)
        pm = np.zeros(int(opm[4]), np.complex64); R.ref_get_prs(mode, pm.ctypes.data, int(opm[4])); out[f"prs_fft_mode{mode}"] = pm
    pi = np.zeros(192, np.uint8); px = np.zeros(6, np.uint8); R.ref_puncture_tables(pi.ctypes.data, px.ctypes.data)
    out["pi_table"] = pi.reshape(24, 8); out["pi_x"] = px
    b = np.zeros(1024, np.uint8); R.ref_scrambler_bytes(b.ctypes.data, 1024); out["prbs_1024"] = b
    # CRC16 known answers
    rng = np.random.default_rng(20251001)
    crc_in = rng.integers(0, 256, (16, 30), dtype=np.uint8)
    out["crc_in"] = crc_in
    out["crc_out"] = np.array([R.ref_crc16(r.ctypes.data, 30) for r in crc_in], dtype=np.uint16)
    # PLL: AVX2+FMA build (the contract) and baseline scalar build
    x = (rng.standard_normal(2552) + 1j * rng.standard_normal(2552)).astype(np.complex64)
    out["pll_in"] = x
    cases = np.array([[0.0, 0.0], [1.7e-3, 0.3], [-2.3e-4, -11.7], [3.6621094e-4, 70.0], [-0.2, 0.49]], dtype=np.float32)
    out["pll_cases"] = cases
    ya = np.empty((len(cases), x.size), np.complex64); yb = np.empty_like(ya)
    for i, (f, dt) in enumerate(cases):
        R.ref_apply_pll(x.ctypes.data, ya[i].ctypes.data, x.size, C.c_float(f), C.c_float(dt))
        R.ref_apply_pll_baseline(x.ctypes.data, yb[i].ctypes.data, x.size, C.c_float(f), C.c_float(dt))
    out["pll_out_avx_fma"] = ya; out["pll_out_baseline"] = yb
    # cyclic prefix correlation (both builds; summation order is build dependent in the reference)
    s = np.zeros((2, 2), np.float32)
    R.ref_conj_mul_sum(x[2048:].ctypes.data, x.ctypes.data, 504, s[0].ctypes.data)
    R.ref_conj_mul_sum_baseline(x[2048:].ctypes.data, x.ctypes.data, 504, s[1].ctypes.data)
    out["cp_corr_avx_fma_and_baseline"] = s
    # scalar chebyshev
    cx = np.linspace(-0.5, 0.5, 257, dtype=np.float32); out["cheb_in"] = cx
    out["cheb_out"] = np.array([R.ref_chebyshev_sine(float(v)) for v in cx], dtype=np.float32)
    # time de-interleaver: 20 CIFs of a 6-CU sub-channel
    n = 6 * 64
    cifs = rng.integers(-127, 128, (20, n), dtype=np.int8); out["deint_in"] = cifs
    h = R.ref_deint_create(n // 8); res = []
    for t in range(20):
        R.ref_deint_consume(h, cifs[t].ctypes.data, n)
        o = np.zeros(n, np.int8); ok = R.ref_deint_deinterleave(h, o.ctypes.data, n)
        res.append(o if ok else np.full(n, -128, np.int8))
    R.ref_deint_destroy(h); out["deint_out"] = np.stack(res)
    # protection-profile plans: all EEP (A/B, levels, several sizes) and all 64 UEP rows
    plans = []
    for tb in (0, 1):
        for lvl in range(4):
            mult = ([12, 8, 6, 4] if tb == 0 else [27, 21, 18, 15])[lvl]
            for nn in (1, 2, 8):
                length = mult * nn
                if length > 864: continue
                pi4 = np.zeros(4, np.int32); lx4 = np.zeros(4, np.int32)
                k = R.ref_subchannel_plan(length, 0, 0, lvl, tb, pi4.ctypes.data, lx4.ctypes.data)
                plans.append([length, 0, 0, lvl, tb, k, *pi4, *lx4])
    for idx in range(64):
        row = np.zeros(4, np.int32); R.ref_uep_row(idx, row.ctypes.data)
        pi4 = np.zeros(4, np.int32); lx4 = np.zeros(4, np.int32)
        k = R.ref_subchannel_plan(int(row[0]), 1, idx, 0, 0, pi4.ctypes.data, lx4.ctypes.data)
        plans.append([int(row[0]), 1, idx, 0, 0, k, *pi4, *lx4])
    out["subchannel_plans"] = np.array(plans, dtype=np.int32)
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
