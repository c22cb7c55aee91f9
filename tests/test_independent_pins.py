"""Independent pins for the rows whose reference translation units cannot be compiled here (dab_viterbi_decoder.cpp needs the absent
vendor/viterbi_decoder, ofdm_demodulator.cpp needs <fftw3.h>).  Nothing below shares code with the oracle or the kernels:

* Viterbi: EXHAUSTIVE maximum-likelihood search over every message of short codewords (every puncturing family
  PI_1..PI_24 + the PI_X tail, every start state) must give the oracle's bytes and its path_error -- this pins the branch metric
  (sum of abs(+-127 - y), punctured symbols as 0), the start metrics (0 / 5080), the trellis orientation and the chain-back bit order.
  A textbook int64 dynamic programme (no u16 wrap, no renormalisation) then pins the renormalisation bookkeeping and the u16 arithmetic
  on FIC- and MSC-length codewords: same bytes, same total error, under either tie rule (strict compare = upstream scalar core,
  min + equality = upstream SIMD cores).  The exhaustive search excludes inputs whose ML path is not unique.
* PRS synchroniser (a5, a6): a float64 numpy model written from ofdm_demodulator.cpp:399-467 and :503-536 must make the same integer
  decisions (peak bin, fast / slow update, fine-time offset, reset) as the oracle over a CFO x timing x SNR grid.
* FFT rounding (a11-a13): a float64 demodulator (numpy FFT) must give the oracle's hard bits on the 20 dB-SNR config-1 frame.
"""
import numpy as np
import pytest

G = (109, 79, 83, 109)          # dab_viterbi_decoder.cpp:25 (decimal form of octal 133,171,145,133)


def parity(v):
    v = np.asarray(v, dtype=np.int64)
    v = v ^ (v >> 4); v = v ^ (v >> 2); v = v ^ (v >> 1)
    return (v & 1).astype(np.int64)


def encode(bits, start_state=0):
    """mother code of a bit sequence (tail included by the caller): register sr = (sr << 1) | bit, outputs parity(sr & G[r])"""
    sr = start_state
    out = []
    for b in bits:
        sr = ((sr << 1) | int(b)) & 0x7F
        out += [int(parity(sr & g)) for g in G]
    return np.array(out, dtype=np.int64)


def kept_mask(segments, n_tail_groups=6):
    """segments = [(kept-count vector of 8 entries, number of 4-symbol groups)]; returns 0/1 per mother symbol"""
    m = []
    for code, groups in segments:
        for g in range(groups):
            k = int(code[g % len(code)])
            m += [1] * k + [0] * (4 - k)
    return np.array(m, dtype=np.int64)


def depuncture(punctured, mask):
    y = np.zeros(mask.size, dtype=np.int64)
    y[mask == 1] = punctured
    return y


def all_messages_cost(y, n_info, start=0):
    """cost[start, message] = 5080*[start != 0] + sum over the n_info + 6 steps of sum_r abs(+-127 - y) along the path of `message`
    from `start` (6 zero tail bits); vectorised over all 64 start states x 2^n_info messages"""
    n_msg = 1 << n_info
    msgs = np.arange(n_msg, dtype=np.int32)[None, :]
    sr = np.repeat(np.arange(64, dtype=np.int32)[:, None], n_msg, axis=1)
    c = np.where(sr == start, 0, 5080).astype(np.int32)           # dab_viterbi_decoder.cpp:36-37, reset(starting_state) :109-112
    par = parity(np.arange(128)[:, None] & np.array(G)[None, :]).astype(np.int32)      # [register, r]
    for t in range(n_info + 6):
        bit = ((msgs >> (n_info - 1 - t)) & 1) if t < n_info else 0
        sr = ((sr << 1) | bit) & 0x7F
        step = np.abs(np.where(par == 1, 127, -127) - y[4 * t:4 * t + 4][None, :]).sum(axis=1).astype(np.int32)   # cost per register value
        c += step[sr]
    return c


@pytest.mark.parametrize("n_info", [8, 16])
def test_viterbi_equals_exhaustive_maximum_likelihood(oracle, n_info):
    rng = np.random.default_rng(100 + n_info)
    pis = range(1, 25) if n_info == 8 else (1, 5, 8, 12, 15, 16, 20, 24)
    checked = 0
    for pi in pis:
        code = oracle.puncture_code(pi)
        tail = oracle.puncture_code_tail()
        mask = kept_mask([(code, n_info), (tail, 6)])
        for trial in range(4 if n_info == 8 else 2):
            msg = rng.integers(0, 2, n_info)
            start = 0 if trial % 2 == 0 else int(rng.integers(1, 64))     # reset(starting_state) with a non-zero state as well
            tx = encode(np.concatenate([msg, np.zeros(6, dtype=np.int64)]), start_state=start)
            sigma = (25.0, 45.0, 70.0, 110.0)[(trial + pi) % 4]            # from clean to mostly wrong
            soft = np.clip(np.rint((2 * tx - 1) * 55.0 + rng.normal(0, sigma, tx.size)), -127, 127).astype(np.int64)
            punct = soft[mask == 1].astype(np.int8)
            y = depuncture(punct, mask)
            costs = all_messages_cost(y, n_info, start)
            best = int(costs.min())
            if int((costs == best).sum()) != 1:
                continue                                                   # ML tie: the winner depends on the tie rule
            si, mi = np.unravel_index(np.argmin(costs), costs.shape)
            n_main = int(mask[:4 * n_info].sum())
            for tie_rule in (0, 1):
                v = oracle.Viterbi(n_info + 6, tie_rule)
                v.reset(start)
                used = v.update(punct[:n_main], code, 4 * n_info)
                used += v.update(punct[n_main:], tail, 24)
                assert used == punct.size
                got, err = v.chainback(n_info // 8, 0)
                exp_bytes = np.packbits(np.array([(int(mi) >> (n_info - 1 - t)) & 1 for t in range(n_info)], dtype=np.uint8))   # MSB first
                # the decoder reports the cost of the best path into state 0 from ANY start state (others penalised by 5080)
                assert err == best, (pi, trial, tie_rule)
                assert np.array_equal(got, exp_bytes), (pi, trial, tie_rule)
                checked += 1
    assert checked >= (100 if n_info == 8 else 20)


def dp_decode(y, n_steps, n_out_bits, tie_rule=0):
    """textbook Viterbi over int64 metrics (no wrap, no renormalisation): returns (bits, total error of the survivor ending in state 0, tie seen)"""
    INF = np.int64(1) << 50
    states = np.arange(64, dtype=np.int64)
    metric = np.full(64, 5080, dtype=np.int64)
    metric[0] = 0
    # transition (old state s, input b) -> new state ((s << 1) | b) & 63 with register sr = ((s << 1) | b) & 127
    prev0 = (states >> 1)                      # predecessors of new state n: n >> 1 and (n >> 1) | 32, input bit = n & 1
    prev1 = (states >> 1) | 32
    sr0 = ((prev0 << 1) | (states & 1)) & 0x7F
    sr1 = ((prev1 << 1) | (states & 1)) & 0x7F
    exp0 = np.stack([np.where(parity(sr0 & g) == 1, 127, -127) for g in G])      # [4, 64]
    exp1 = np.stack([np.where(parity(sr1 & g) == 1, 127, -127) for g in G])
    choice = np.zeros((n_steps, 64), dtype=np.int8)
    tie = False
    for t in range(n_steps):
        sym = y[4 * t:4 * t + 4][:, None]
        c0 = metric[prev0] + np.abs(exp0 - sym).sum(axis=0)
        c1 = metric[prev1] + np.abs(exp1 - sym).sum(axis=0)
        tie |= bool(np.any(c0 == c1))
        choice[t] = (c1 <= c0) if tie_rule else (c1 < c0)        # the two upstream cores: strict compare / min + equality
        metric = np.minimum(c0, c1)
    s = 0
    bits = np.zeros(n_steps, dtype=np.uint8)
    for t in range(n_steps - 1, -1, -1):
        bits[t] = s & 1
        s = ((s >> 1) | 32) if choice[t, s] else (s >> 1)
    return bits[:n_out_bits], int(metric[0]), tie


@pytest.mark.parametrize("shape", ["fic", "msc_eep3a", "long_weak"])
def test_viterbi_equals_int64_dynamic_programme_on_long_codewords(oracle, shape):
    """FIC (774 steps) and 48 CU EEP 3-A (1542 steps) with heavy noise: the u16 metrics renormalise several times; bytes and the
    reported total error must equal the wrap-free int64 decoder"""
    rng = np.random.default_rng({"fic": 1, "msc_eep3a": 2, "long_weak": 3}[shape])
    segs = {"fic": [(16, 21 * 32), (15, 3 * 32)], "msc_eep3a": [(8, 45 * 32), (7, 3 * 32)], "long_weak": [(24, 40 * 32), (1, 24 * 32)]}[shape]
    n_info = sum(g for _, g in segs)
    tail = oracle.puncture_code_tail()
    mask = kept_mask([(oracle.puncture_code(pi), g) for pi, g in segs] + [(tail, 6)])
    done = 0
    for trial in range(6):
        msg = rng.integers(0, 2, n_info)
        tx = encode(np.concatenate([msg, np.zeros(6, dtype=np.int64)]))
        gain, sigma = ((60.0, 50.0), (100.0, 60.0), (30.0, 70.0))[trial % 3]
        soft = np.clip(np.rint((2 * tx - 1) * gain + rng.normal(0, sigma, tx.size)), -127, 127).astype(np.int64)
        punct = soft[mask == 1].astype(np.int8)
        y = depuncture(punct, mask)
        for tie_rule in (0, 1):
            bits, total, tie = dp_decode(y, n_info + 6, n_info, tie_rule)
            v = oracle.Viterbi(n_info + 6, tie_rule)
            v.reset(0)
            pos = 0
            for pi, g in segs:
                pos += v.update(punct[pos:], oracle.puncture_code(pi), 4 * g)
            pos += v.update(punct[pos:], tail, 24)
            assert pos == punct.size
            got, err = v.chainback(n_info // 8, 0)
            assert err == total, (shape, trial, tie_rule)              # the ML cost does not depend on how ties are broken
            assert np.array_equal(got, np.packbits(bits)), (shape, trial, tie_rule)
            done += 1
    assert done == 12


def dp_decode_core(y, n_steps, n_out_bits, core):
    """the two upstream cores as MODELS of their published add-compare-select, over Python integers: core 0 = ViterbiDecoder_Scalar (uint16_t sums
    that wrap, `m0 > m1`), core 1 = the SIMD cores (adds_epu16: the sums saturate at 65535; min_epu16; decision = cmpeq(survivor, upper candidate));
    both with the reference's renormalisation (dab_viterbi_decoder.cpp:31-41).  Returns (bits, path error, largest candidate sum seen before reduction)"""
    metric = [5080] * 64
    metric[0] = 0
    total, biggest = 0, 0
    choice = np.zeros((n_steps, 64), dtype=np.int8)
    par = [[int(parity(sr & g)) for g in G] for sr in range(128)]
    for t in range(n_steps):
        new = [0] * 64
        for n in range(64):
            cand = []
            for p in (n >> 1, (n >> 1) | 32):
                sr = ((p << 1) | (n & 1)) & 0x7F
                e = sum(abs((127 if par[sr][r] else -127) - int(y[4 * t + r])) for r in range(4))
                v = metric[p] + e
                biggest = max(biggest, v)
                cand.append(min(v, 65535) if core else (v & 0xFFFF))
            up = (cand[1] <= cand[0]) if core else (cand[1] < cand[0])
            choice[t, n] = up
            new[n] = cand[1] if up else cand[0]
        if new[0] >= 60455:
            mn = min(new)
            new = [v - mn for v in new]
            total += mn
        metric = new
    s = 0
    bits = np.zeros(n_steps, dtype=np.uint8)
    for t in range(n_steps - 1, -1, -1):
        bits[t] = s & 1
        s = ((s >> 1) | 32) if choice[t, s] else (s >> 1)
    return bits[:n_out_bits], total + metric[0], biggest


def test_u16_candidate_sums_cannot_reach_65535():
    """Where the two upstream cores could differ besides ties: the scalar core's uint16_t sums wrap, the SIMD cores' adds_epu16 saturates.  For THIS code
    (K = 7, polynomials 133 171 145 133) and symbols in [-127, 127] (include/dabgpu.h: -128 is read as -127) no input makes a sum reach 65535, so the two models differ in the tie rule only:
      * after the first six steps every state has a path from every state of six steps ago, hence  metric(s) - min metric <= 254 * d(s ^ s*), d = the
        weight of the lightest path from the all-zero state to that state difference (the code is linear) -- the fixed point computed below: 15 bits;
      * a candidate adds one branch (<= 4 x 254; the complementary cost 1016 - e of the butterfly stays in [0, 1016] too), and the reference renormalises as soon as metric[0] >= 60455 (dab_viterbi_decoder.cpp:31-41), so before
        any step min metric <= metric[0] <= 60454:  candidate <= 60454 + 254 * (15 + 4) = 65280 < 65535;
      * in the first six steps after reset() the level is at most 5080 + 6 * 1016.
    The bound is computed from the polynomials here (nothing shared with the oracle or the kernels) and then attacked: the hardest inputs a hill climb
    finds (erasures up to just below the renormalisation threshold, then 15-17 steps of extreme symbols) stay below it, and both core models -- and the
    oracle under both -- decode them identically."""
    d = [10 ** 9] * 64
    d[0] = 0
    for _ in range(64):
        nd = [min(d[p] + int(sum(parity((((p << 1) | (n & 1)) & 0x7F) & g) for g in G)) for p in (n >> 1, (n >> 1) | 32)) for n in range(64)]
        nd[0] = 0
        if nd == d:
            break
        d = nd
    spread_bits = max(d)
    assert spread_bits == 15
    bound = 60454 + 254 * (spread_bits + 4)
    assert bound == 65280 and bound < 65535
    import oracle as O
    O.build()
    rng = np.random.default_rng(5)
    worst = 0
    for n_a in (117, 118, 119):                                   # 508 per erased step: 59436, 59944, 60452 -- the last is 3 below the threshold
        n_steps = 134                                              # 128 message bits + the 6 tail steps the chain-back skips
        n_b = n_steps - n_a
        y = np.concatenate([np.zeros(4 * n_a, np.int64), rng.choice([-127, 127], size=4 * n_b)])
        _, _, best = dp_decode_core(y, n_steps, 8, 0)
        for _ in range(60):                                        # hill climb on the extreme symbols
            y2 = y.copy()
            y2[4 * n_a + rng.integers(0, 4 * n_b)] = rng.choice([-127, 127, 0])
            b2 = dp_decode_core(y2, n_steps, 8, 0)[2]
            if b2 >= best:
                y, best = y2, b2
        worst = max(worst, best)
        n_bits = n_steps - 6
        r0, r1 = dp_decode_core(y, n_steps, n_bits, 0), dp_decode_core(y, n_steps, n_bits, 1)
        assert r0[2] == r1[2] <= bound
        for core, r in ((0, r0), (1, r1)):
            v = O.Viterbi(n_steps, core)
            v.reset(0)
            assert v.update(y.astype(np.int8), np.array([4], np.uint8), 4 * n_steps) == 4 * n_steps
            got, err = v.chainback(n_bits // 8, 0)
            assert err == r[1] and np.array_equal(got, np.packbits(r[0])), (n_a, core)
    assert 60454 < worst <= bound, worst                          # the attack does get past the threshold level, and not past the bound


# ---------------------------------------------------------------------------------------------------------------------
# PRS synchroniser: float64 model of ofdm_demodulator.cpp:360-548
# ---------------------------------------------------------------------------------------------------------------------
def f64_coarse(prs_sym, prs_fft, coarse, found, cfg):
    n = 2048
    X = np.fft.fft(prs_sym[:n].astype(np.complex128))
    rel = np.conj(X) * np.roll(X, -1)                                   # CalculateRelativePhase :901-909: arg(conj(z0) z1)
    rel[-1] = 0
    ref_rel = np.conj(prs_fft.astype(np.complex128)) * np.roll(prs_fft.astype(np.complex128), -1)
    ref_rel[-1] = 0
    tref = np.conj(np.fft.ifft(ref_rel))                                # constructor :127-135
    corr = np.fft.fft(np.fft.ifft(rel) * tref)
    mag = 20.0 * np.log10(np.abs(np.fft.fftshift(corr)) + 1e-300)       # CalculateMagnitude :911-920 (fft-shifted)
    M = n // 2
    mco = min(max(int(cfg.max_coarse_freq_correction_norm * n), 0), M)
    idx = [i for i in range(-mco, mco + 1) if i + M != n]
    vals = np.array([mag[i + M] for i in idx])
    k = int(np.argmax(vals))
    srt = np.sort(vals)
    margin = srt[-1] - srt[-2]
    max_index = idx[k]

    def peak(index):
        index = min(max(index, -mco), mco)
        fi = min(index + M, n - 1)
        return fi - M, 10.0 ** (mag[fi] / 20.0)
    pk = [peak(max_index - 1), peak(max_index), peak(max_index + 1)]
    s = sum(p[1] for p in pk)
    lerp = sum(p[0] * p[1] / s for p in pk)
    pred = -lerp / n
    err = pred - coarse
    large = abs(err) > 1.5 / n
    fast = large or not found
    return max_index, fast, pred, margin, abs(abs(err) - 1.5 / n)


def f64_fine(prs_sym, prs_fft, freq, cfg):
    n, cp, period = 2048, 504, 2552
    x = prs_sym[:n].astype(np.complex128) * np.exp(2j * np.pi * freq * np.arange(n))
    imp = n * np.fft.ifft(np.fft.fft(x) * np.conj(prs_fft.astype(np.complex128)))   # FFTW's backward transform is unnormalised, and the
    db = 20.0 * np.log10(np.abs(imp) + 1e-300)                                      # distance weighting below multiplies dB VALUES: scale matters
    w = 1.0 - (1.0 - cfg.impulse_peak_distance_probability) * np.abs(cp - np.arange(n)) / period
    weighted = w * db
    best, bi = db[0], 0                                                  # :503 initialised with the UNWEIGHTED [0]
    for i in range(n):
        if weighted[i] > best:
            best, bi = weighted[i], i
    srt = np.sort(weighted)
    avg = db.mean()
    ok = (best - avg) >= cfg.impulse_peak_threshold_db
    return ok, bi - cp, srt[-1] - srt[-2], abs((best - avg) - cfg.impulse_peak_threshold_db)


def test_sync_decisions_equal_a_float64_model(oracle):
    rng = np.random.default_rng(77)
    prs_fft = oracle.prs_fft()
    conj_ref, time_ref = oracle.sync_refs()
    cfg = oracle.sync_cfg_default()
    bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
    tx0 = np.concatenate([oracle.modulate_frame(bits)] * 2)
    agreed = resets = 0
    for cfo_bins in (0.0, 0.3, 0.5, 1.0, 2.6, -3.7, 11.25, -40.49, 200.2, -333.0):
        tx_c = oracle.apply_pll(tx0, cfo_bins / 2048.0, 0.1)
        for toff in (0, 1, 37, -80, 200, -250):
            for noise in (0.0, 4.0, 20.0, 60.0):                         # PRS samples have |x| ~ 39: SNR from clean to -4 dB
                tx = tx_c
                if noise > 0:
                    tx = (tx_c + noise * (rng.standard_normal(tx_c.size) + 1j * rng.standard_normal(tx_c.size))).astype(np.complex64)
                start = oracle.NB_NULL_PERIOD - toff
                sym = tx[start:start + 2048].copy()
                st = oracle.SyncState(0.0, 0.0, 0, 0, 0, 0)
                coarse64, found64 = 0.0, False
                for it in range(2):                                      # first estimate (fast), then the tracking call
                    before = float(st.freq_coarse)
                    fr = oracle.coarse_freq_sync(sym, st, cfg, time_ref)
                    max_index, fast, pred, margin, thr_margin = f64_coarse(sym, prs_fft, coarse64, found64, cfg)
                    if margin < 0.02 or thr_margin < 2e-6:               # float32 vs float64 may legitimately order a near-tie differently
                        break
                    o_peak = int(np.argmax(fr[1024 - 1024:1024 + 1024 + 0])) - 1024     # oracle's dB response, fft-shifted
                    assert o_peak == max_index, (cfo_bins, toff, noise, it)
                    delta = float(st.freq_coarse) - before
                    beta = 1.0 if fast else cfg.coarse_freq_slow_beta
                    assert abs(delta - beta * (pred - coarse64)) < 3e-6, (cfo_bins, toff, noise, it, fast)
                    coarse64 += beta * (pred - coarse64)
                    found64 = True
                    f = float(np.float32(st.freq_coarse) + np.float32(st.freq_fine))
                    ok_o, off_o, ir = oracle.fine_time_sync(sym, f, cfg, conj_ref)
                    ok64, off64, m2, thr2 = f64_fine(sym, prs_fft, f, cfg)
                    if thr2 < 0.05 or (ok64 and m2 < 0.02):
                        break
                    assert ok_o == ok64, (cfo_bins, toff, noise, it)
                    if ok64:
                        assert off_o == off64, (cfo_bins, toff, noise, it)
                        agreed += 1
                    else:
                        resets += 1
    assert agreed > 250 and resets > 10


def test_float64_demodulator_gives_the_same_hard_bits_on_the_config1_frame(oracle):
    """CFO +3.7 bins, AWGN 20 dB SNR, seed 1, u8 quantised; the oracle's own FFT (float32, 4x8x8x8) against numpy's float64"""
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
    cfo = 3.7 / 2048.0
    tx = oracle.apply_pll(oracle.modulate_frame(bits), cfo, 0.0)
    sig = np.sqrt(np.mean(np.abs(tx[oracle.NB_NULL_PERIOD:]) ** 2))
    tx = tx + (sig / 10.0 / np.sqrt(2.0)) * (rng.standard_normal(tx.size) + 1j * rng.standard_normal(tx.size))
    scale = 127.5 / (4.0 * sig)                                          # u8 capture with ~4 sigma head-room (app_iq_readers.h:23-30 inverse)
    q = np.clip(np.rint(np.stack([tx.real, tx.imag], -1) * scale + 127.5), 0, 255).astype(np.uint8)
    iq = ((q.astype(np.float32) - 127.5) / 127.5)
    x = (iq[:, 0] + 1j * iq[:, 1]).astype(np.complex64)
    frame = oracle.tx_to_frame_buffer(x)
    got = oracle.demod_frame(frame, -cfo)["bits"]
    # float64 model of steps 5-8 of SURVEY A.2
    n = np.arange(196608)
    y = frame.astype(np.complex128) * np.exp(2j * np.pi * (-cfo) * n)
    syms = y[:76 * 2552].reshape(76, 2552)[:, 504:]
    X = np.fft.fft(syms, axis=1)
    m = oracle.mapper().astype(np.int64)
    carriers = np.concatenate([np.arange(2048 - 768, 2048), np.arange(1, 769)])
    d = X[:-1, carriers] * np.conj(X[1:, carriers])                      # X_i conj(X_{i+1}), natural carrier order
    v = d[:, m]
    A = np.maximum(np.abs(v.real), np.abs(v.imag))
    soft = np.concatenate([-(v.real / A) * 127.0, (v.imag / A) * 127.0], axis=1).reshape(-1)
    hard64 = soft >= 0
    hard = got.astype(np.int32) >= 0
    # identical wherever the float64 value is not within rounding distance of the decision threshold
    safe = np.abs(soft) > 1.5
    assert np.array_equal(hard[safe], hard64[safe])
    assert safe.mean() > 0.98
    assert (hard != bits.astype(bool)).mean() < 5e-3                     # and both are the transmitted bits but for channel errors
