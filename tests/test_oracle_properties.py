"""Known-answer and round-trip properties of the oracle for the parts of the path whose reference code
cannot be compiled here (FFTW and vendor/viterbi_decoder are absent): DFT known answers, TX->RX
loop-back, conv-encode -> Viterbi round trips, FIB CRC16."""
import numpy as np
import pytest


def test_fft_matches_dft(oracle):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    X = oracle.fft2048(x)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(X - ref).max() <= 4e-7 * np.abs(ref).max() * np.sqrt(11)
    Xi = oracle.fft2048(x, inverse=True)
    refi = np.fft.ifft(x.astype(np.complex128)) * 2048
    assert np.abs(Xi - refi).max() <= 4e-7 * np.abs(refi).max() * np.sqrt(11)
    # impulses and single tones are exact up to twiddle rounding
    for k in (0, 1, 5, 1023, 2047):
        e = np.zeros(2048, np.complex64); e[k] = 1
        assert np.allclose(oracle.fft2048(e), np.exp(-2j * np.pi * k * np.arange(2048) / 2048), atol=2e-7)
    # linearity / Parseval
    assert np.isclose(np.sum(np.abs(X) ** 2), 2048 * np.sum(np.abs(x) ** 2), rtol=1e-5)


def test_reference_modulator_loopback(oracle):
    """bits expected from OFDM_Modulator's PHASE_MAP (ofdm_modulator.cpp:102-126) after the receiver's
    de-interleave (ofdm_demodulator.cpp:874)"""
    rng = np.random.default_rng(11)
    pay = rng.integers(0, 256, 75 * 384, dtype=np.uint8)
    tx = oracle.modulate_frame_reference_payload(pay)
    r = oracle.demod_frame(oracle.tx_to_frame_buffer(tx), 0.0)
    m = oracle.mapper()
    d = pay.reshape(75, 384)
    dib = np.stack([(d >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(75, 1536)
    hre = np.isin(dib, [0, 3]).astype(np.uint8)      # Re<0 -> logical 1
    him = np.isin(dib, [0, 1]).astype(np.uint8)      # Im<0 -> logical 1
    exp = np.concatenate([hre[:, m], him[:, m]], axis=1).reshape(-1)
    assert np.array_equal((r["bits"] >= 0).astype(np.uint8), exp)
    assert np.all(np.abs(r["bits"].astype(np.int16)) >= 126)    # noise-free: |soft| saturates on both axes
    assert abs(float(r["total_phase"])) < 1e-3


def test_frequency_offset_loop(oracle):
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
    tx = oracle.modulate_frame(bits)
    f = 0.3 / 2048
    fb = oracle.tx_to_frame_buffer(oracle.apply_pll(tx, f, 0.0))
    r = oracle.demod_frame(fb, 0.0)
    assert np.isclose(float(r["total_phase"]), 76 * 2 * np.pi * f * 2048, rtol=1e-3)
    fine = 0.0
    for _ in range(12):                                # IIR converges to -f (ofdm_demodulator.cpp:615-618)
        r = oracle.demod_frame(fb, fine)
        fine = float(oracle.update_fine_freq(fine, r["total_phase"]))
    assert abs(fine + f) < 2e-8
    assert np.array_equal((r["bits"] >= 0).astype(np.uint8), bits)


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_fic_round_trip(oracle, tie_rule):
    rng = np.random.default_rng(21 + tie_rule)
    fib = rng.integers(0, 256, 90, dtype=np.uint8)
    enc = oracle.fic_encode_group(fib)
    soft = oracle.soft_from_bits(enc)
    out, mask, err = oracle.fic_decode_group(soft, tie_rule)
    assert mask == 7 and np.array_equal(out.reshape(3, 32)[:, :30].reshape(-1), fib)
    assert err == 127 * (3096 - 2304)                  # punctured positions cost 127 each on the true path
    noisy = np.clip(soft.astype(np.float32) * 0.5 + rng.standard_normal(soft.size) * 25, -127, 127).astype(np.int8)
    out, mask, err = oracle.fic_decode_group(noisy, tie_rule)
    assert mask == 7 and np.array_equal(out.reshape(3, 32)[:, :30].reshape(-1), fib)
    # a corrupted FIB must fail its CRC (fic_decoder.cpp:113-115 drops it)
    bad = soft.copy(); bad[:700] = -bad[:700]
    out, mask, _ = oracle.fic_decode_group(bad, tie_rule)
    assert (mask & 1) == 0


@pytest.mark.parametrize("profile", [
    dict(length=48, eep_level=2, eep_type=0),      # EEP 3-A, 64 kbps: canonical multiplex sub-channel
    dict(length=8, eep_level=1, eep_type=0),       # EEP 2-A special case n=1
    dict(length=27, eep_level=0, eep_type=1),      # EEP 1-B
    dict(length=16, is_uep=True, uep_index=0),     # UEP 32 kbps level 5 (3 segments, L4 = 0)
    dict(length=35, is_uep=True, uep_index=4),     # UEP with 4 padding bits
])
def test_msc_round_trip_through_time_interleaver(oracle, profile):
    rng = np.random.default_rng(99)
    sc = oracle.subchannel(0, **profile)
    _, _, nb = oracle.subchannel_plan(sc)
    T = 19
    data = [rng.integers(0, 256, nb, dtype=np.uint8) for _ in range(T)]
    lf = np.stack([oracle.msc_encode_logical(sc, d) for d in data])
    cifs = oracle.time_interleave(lf)
    D = oracle.Deinterleaver(sc.length * 8)
    n_ok = 0
    for t in range(T):
        D.consume(oracle.soft_from_bits(cifs[t]))
        o = D.deinterleave()
        assert (o is None) == (t < 15)                 # cif_deinterleaver.cpp:40-42
        if o is not None:
            dec, _ = oracle.msc_decode_logical(sc, o)
            assert np.array_equal(dec, data[t - 15])
            n_ok += 1
    assert n_ok == T - 15


def test_viterbi_decisions_layout(oracle):
    """decision word bit j = survivor choice of new state j; chainback emits MSB-first bytes"""
    rng = np.random.default_rng(2)
    data = rng.integers(0, 256, 16, dtype=np.uint8)
    mother = oracle.conv_encode(data)
    v = oracle.Viterbi(128)
    v.reset()
    n = v.update(oracle.soft_from_bits(mother), np.array([4] * 8, np.uint8), mother.size)
    assert n == mother.size and v.decoded_bits() == 128 + 6
    out, err = v.chainback(16)
    assert np.array_equal(out, data) and err == 0
    assert v.metrics()[0] == 0


def test_avx2_trellis_step_of_the_oracle_equals_its_scalar_statement(tmp_path):
    """The oracle's add-compare-select runs sixteen butterflies per vector where the CPU has AVX2 + BMI2 (what an upstream build does with
    its SIMD core; otherwise bench.py's cpu_baseline_full would time a scalar port); DAB_ORACLE_SCALAR_VITERBI=1 selects the scalar
    statement.  Decoded bytes, path errors, final metrics, every decision word and the consumed counts must be identical for both tie
    rules -- on noisy code words, on -128 inputs (branch errors above 1016: the u16 complement wraps), on all-zero input (every compare a
    tie), on weak signals (frequent renormalisation), from random start and end states."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys
sys.path.insert(0, %r)
import numpy as np, oracle as O
rng = np.random.default_rng(int(sys.argv[2]))
outs = []
for tie in (0, 1):
    for case in range(8):
        n_bytes = [96, 192, 24, 576, 96, 96, 300, 48][case]
        steps = n_bytes * 8 + 6
        amp = [127, 127, 40, 127, 127, 5, 127, 127][case]
        if case == 4:
            soft = np.full(4 * steps, -128, np.int8)
        elif case == 5:
            soft = rng.integers(-amp, amp + 1, 4 * steps).astype(np.int8)
        elif case == 6:
            soft = np.zeros(4 * steps, np.int8)
        else:
            soft = (rng.integers(0, 2, 4 * steps) * 2 - 1).astype(np.int8) * amp
            soft = np.clip(soft.astype(np.int32) + rng.integers(-90, 91, soft.size), -128, 127).astype(np.int8)
        v = O.Viterbi(n_bytes * 8, tie)
        v.reset(int(rng.integers(0, 64)))
        used = v.update(soft, np.array([4], np.uint8), 4 * steps)
        by, err = v.chainback(n_bytes, int(rng.integers(0, 64)))
        outs.append(np.concatenate([by, np.frombuffer(np.uint64(err).tobytes(), np.uint8), v.metrics().view(np.uint8), v.decisions(steps).view(np.uint8),
                                    np.frombuffer(np.uint64(used).tobytes(), np.uint8)]))
np.save(sys.argv[1], np.concatenate(outs))
''' % os.path.join(root, "oracle")
    for seed in (1, 2):
        got = []
        for tag, env in (("vector", {}), ("scalar", {"DAB_ORACLE_SCALAR_VITERBI": "1"})):
            out = tmp_path / f"{tag}_{seed}.npy"
            subprocess.run([sys.executable, "-c", code, str(out), str(seed)], env=dict(os.environ, **env), check=True, timeout=300)
            got.append(np.load(out))
        assert got[0].size > 100000 and np.array_equal(got[0], got[1]), seed
