"""CPU pins of the DAB+ outer-code oracle (oracle/dab_oracle_dabplus.c) against the reference's own objects compiled in
place (Reed_Solomon_Decoder, AAC_Frame_Processor -> oracle/_ref) and against the golden vectors those objects produced
(tests/golden/dabplus_vectors.npz, generator tests/golden/make_golden_dabplus.py)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gdp():
    return np.load(os.path.join(ROOT, "tests", "golden", "dabplus_vectors.npz"))


def test_rs_matches_reference_vectors(oracle, gdp):
    cws, counts, fixed, positions = gdp["rs_in"], gdp["rs_count"], gdp["rs_out"], gdp["rs_positions"]
    assert set(np.unique(counts)) >= {-1, 0, 1, 2, 3, 4, 5}
    for k in range(cws.shape[0]):
        n, out, pos = oracle.rs120_decode(cws[k])
        assert n == counts[k], k
        assert np.array_equal(out, fixed[k]), k
        assert np.array_equal(pos, positions[k][:max(n, 0)]), k


def test_rs_round_trip_and_capacity(oracle):
    rng = np.random.default_rng(1)
    for e in range(0, 6):
        data = rng.integers(0, 256, 110, dtype=np.uint8)
        cw = np.concatenate([data, oracle.rs120_encode(data)])
        assert oracle.rs120_decode(cw)[0] == 0
        bad = cw.copy()
        where = rng.choice(120, e, replace=False)
        for j in where:
            bad[j] ^= rng.integers(1, 256)
        n, out, pos = oracle.rs120_decode(bad)
        assert n == e and np.array_equal(out, cw)
        assert sorted(int(p) - 135 for p in pos) == sorted(int(j) for j in where)


def test_superframe_sequences_match_reference_vectors(oracle, gdp):
    import dabplus_model  # noqa: F401  (documents where the sequences come from)
    for name in [str(s) for s in gdp["seq_names"]]:
        frames, n = gdp[f"{name}_frames"], int(gdp[f"{name}_n"])
        exp = gdp[f"{name}_ref"]                      # per frame: the 12 reference event fields
        exp_len, exp_bytes = gdp[f"{name}_au_len"], gdp[f"{name}_au_bytes"]
        p = oracle.AacFrameProcessor()
        for k in range(frames.shape[0]):
            rc, r, sf = p.process(frames[k][:n])
            e = exp[k]
            check_against_reference_events(r, sf, e, exp_len[k], exp_bytes[k], (name, k))


def check_against_reference_events(r, sf, e, au_len, au_bytes, where):
    """r: oracle record, e: {firecode_errors, rs_failed_index, header_valid, rate, ps, sbr, stereo, surround, num_aus, ok, bad}"""
    fire_events = int(r["firecode_wait_failed"]) + int(bool(r["superframe_done"]) and r["rs_failed_index"] < 0 and not r["firecode_ok"])
    assert fire_events == e[0], where
    assert int(r["rs_failed_index"]) == e[1], where
    assert int(r["header_valid"]) == e[2], where
    if e[2]:
        d = int(r["descriptor"])
        assert (48000 if d & 0x40 else 32000) == e[3] and ((d >> 3) & 1) == e[4] and ((d >> 5) & 1) == e[5] and ((d >> 4) & 1) == e[6], where
        ok = int(r["au_crc_ok_mask"])
        assert ok == int(np.uint32(e[9])), where
        stop = int(r["au_walk_stopped_at"])
        walked = int(r["num_aus"]) if stop < 0 else stop
        assert (((1 << walked) - 1) & ~ok) == int(np.uint32(e[10])), where
        for i in range(6):
            if ok & (1 << i):
                a, b = int(r["au_start"][i]), int(r["au_start"][i + 1]) - 2
                assert b - a == au_len[i] and np.array_equal(sf[a:b], au_bytes[i][:b - a]), (where, i)


def test_superframes_against_reference_objects_in_place(oracle):
    """wider randomised sweep, only where oracle/_ref was built"""
    import ctypes as C
    import dabplus_model as M
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_aac_create"):
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(11)
    for n in (24, 96, 192, 264):
        h = C.c_void_p(R.ref_aac_create())
        p = oracle.AacFrameProcessor()
        stream = []
        for s in range(14):
            sf, _, _ = M.make_superframe(oracle, rng, n, dac_rate=int(rng.integers(0, 2)), sbr_flag=int(rng.integers(0, 2)),
                                         bad_au_crc=(1,) if s % 5 == 4 else ())
            errs = [0, 0, 1, 3, 5, 6, 0, 8, 2, 0, 5, 0, 7, 0][s]
            sf = M.corrupt(rng, sf, {int(rng.integers(0, sf.size // 120)): errs})
            stream += list(sf.reshape(5, n))
        stream = stream[3:]                                      # start in the middle of a super frame
        for k, fr in enumerate(stream):
            out12 = np.zeros(12, np.int32); au_len = np.zeros(6, np.int32); au_bytes = np.zeros((6, 2048), np.uint8)
            R.ref_aac_process(h, fr.ctypes.data, n, out12.ctypes.data, au_len.ctypes.data, au_bytes.ctypes.data, 2048)
            rc, r, sf_o = p.process(fr)
            check_against_reference_events(r, sf_o, out12, au_len, au_bytes, (n, k))
        R.ref_aac_destroy(h)
    # Reed-Solomon alone, including words far beyond the correction capacity
    for k in range(3000):
        data = rng.integers(0, 256, 110, dtype=np.uint8)
        cw = np.concatenate([data, oracle.rs120_encode(data)])
        e = int(rng.integers(0, 12))
        for j in rng.choice(120, e, replace=False):
            cw[j] ^= rng.integers(1, 256)
        a = cw.copy(); pos = np.full(10, -1, np.int32)
        nref = R.ref_rs120_decode(a.ctypes.data, pos.ctypes.data)
        n, out, p_o = oracle.rs120_decode(cw)
        assert n == nref and np.array_equal(out, a) and np.array_equal(p_o, pos[:max(n, 0)]), (k, e)
