"""GPU parity tests of the Viterbi / FIC / MSC path through the C ABI against the CPU oracle.
Bar: decoded bytes, FIB CRC masks and path errors BIT-EXACT (integer work), for both tie-break rules, on
noisy inputs (so that survivor decisions are exercised), including the renormalisation branch, puncture
boundaries, the time de-interleaver ring and edge cases (all-erased input, -128 soft bits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


def noisy(oracle, bits01, rng, gain=0.45, sigma=30.0):
    s = oracle.soft_from_bits(bits01).astype(np.float32) * gain + rng.standard_normal(len(bits01)) * sigma
    return np.clip(np.rint(s), -127, 127).astype(np.int8)


def results_np(t):
    import dabgpu
    return t.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(-1)


@pytest.fixture(params=[1, 2, 3], ids=["wave", "lane", "octet"])
def mapping(ctx, request):
    """the three device mappings of the decoder (include/dabgpu.h DABGPU_VIT_MAP_*), forced"""
    ctx.viterbi_set_mapping(request.param)
    yield request.param
    ctx.viterbi_set_mapping(0)


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_fic_frames_match_oracle(ctx, oracle, tie_rule, mapping):
    import torch
    rng = np.random.default_rng(100 + tie_rule)
    n_frames = 9
    frames = np.zeros((n_frames, oracle.NB_FRAME_BITS), dtype=np.int8)
    for f in range(n_frames):
        for g in range(4):
            enc = oracle.fic_encode_group(rng.integers(0, 256, 90, dtype=np.uint8))
            sigma = [0.0, 20.0, 35.0, 60.0][(f + g) % 4]          # from clean to mostly-failing CRCs
            frames[f, g * 2304:(g + 1) * 2304] = noisy(oracle, enc, rng, sigma=sigma) if sigma else oracle.soft_from_bits(enc)
    frames[:, 9216:] = rng.integers(-127, 128, (n_frames, oracle.NB_FRAME_BITS - 9216), dtype=np.int8)
    d_bits = torch.from_numpy(frames).cuda()
    d_out = torch.zeros((n_frames, 4, 96), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((n_frames * 4, 16), dtype=torch.uint8, device="cuda")
    ctx.fic_decode_frames(d_bits, n_frames, d_out, d_res, tie_rule=tie_rule)
    torch.cuda.synchronize()
    out, res = d_out.cpu().numpy(), results_np(d_res)
    n_fail = 0
    for f in range(n_frames):
        for g in range(4):
            eb, em, ee = oracle.fic_decode_group(frames[f, g * 2304:(g + 1) * 2304], tie_rule)
            r = res[f * 4 + g]
            assert np.array_equal(out[f, g], eb), (f, g)
            assert int(r["crc_ok_mask"]) == em and int(r["path_error"]) == ee and int(r["n_out_bytes"]) == 96
            n_fail += (em != 7)
    assert 0 < n_fail < n_frames * 4, "noise levels must produce both passing and failing FIB CRCs"


def test_generic_batch_direct_mode_all_puncture_vectors(ctx, oracle):
    """every PI_1..PI_24 as a single segment, random lengths, noisy; renormalisation must trigger on the long ones"""
    import dabgpu
    import torch
    rng = np.random.default_rng(7)
    cws, expect, bufs = [], [], []
    for pi in range(1, 25):
        L = int(rng.integers(1, 7)) if pi % 5 else 60           # 60 blocks = 1920 steps: enough to renormalise
        n_bits = 32 * L
        data = rng.integers(0, 256, n_bits // 8, dtype=np.uint8)
        mother = oracle.conv_encode(data)
        code = oracle.puncture_code(pi)
        kept = np.empty(mother.size, np.uint8)
        n1 = oracle.lib().dab_puncture(mother.ctypes.data, 128 * L, code.ctypes.data, 8, kept.ctypes.data)
        tail = np.empty(24, np.uint8)
        tail_in = np.ascontiguousarray(mother[128 * L:])
        n2 = oracle.lib().dab_puncture(tail_in.ctypes.data, 24, oracle.puncture_code_tail().ctypes.data, 6, tail.ctypes.data)
        enc = np.concatenate([kept[:n1], tail[:n2]])
        soft = noisy(oracle, enc, rng, gain=0.5, sigma=22.0 + pi)
        v = oracle.Viterbi(n_bits, 0)
        v.reset()
        used = v.update(soft, code, 128 * L)
        used += v.update(soft[used:], oracle.puncture_code_tail(), 24)
        assert used == soft.size
        ob, oe = v.chainback(n_bits // 8)
        ob ^= oracle.scrambler_bytes(n_bits // 8)
        d_in = torch.from_numpy(soft).cuda()
        d_o = torch.zeros(n_bits // 8, dtype=torch.uint8, device="cuda")
        bufs.append((d_in, d_o))
        cw = dabgpu.Codeword()
        cw.d_src, cw.d_out, cw.n_steps = d_in.data_ptr(), d_o.data_ptr(), n_bits + 6
        cw.seg_pi[0], cw.seg_steps[0] = pi, 32 * L
        cws.append(cw)
        expect.append((ob, oe))
    d_res = torch.zeros((len(cws), 16), dtype=torch.uint8, device="cuda")
    ctx.viterbi_decode_batch(cws, d_res, tie_rule=0)
    torch.cuda.synchronize()
    res = results_np(d_res)
    saw_renorm = False
    for i, (ob, oe) in enumerate(expect):
        assert np.array_equal(bufs[i][1].cpu().numpy(), ob), f"PI_{i + 1}"
        assert int(res[i]["path_error"]) == oe, f"PI_{i + 1}"
        saw_renorm |= oe > 65535
    assert saw_renorm, "no codeword was long/noisy enough to pass the renormalisation threshold"


def test_edge_inputs(ctx, oracle, mapping):
    """all-erased codeword (every metric ties), -128 soft bits (read as -127), rejected descriptors"""
    import dabgpu
    import torch
    rng = np.random.default_rng(9)
    zeros = np.zeros(2304, np.int8)
    enc = oracle.fic_encode_group(rng.integers(0, 256, 90, dtype=np.uint8))
    sat = np.where(enc > 0, 127, -128).astype(np.int8)
    frames = np.zeros((2, oracle.NB_FRAME_BITS), np.int8)
    frames[0, :2304] = zeros
    frames[1, :2304] = sat
    d_bits = torch.from_numpy(frames).cuda()
    d_out = torch.zeros((2, 4, 96), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((8, 16), dtype=torch.uint8, device="cuda")
    for tie in (0, 1):
        ctx.fic_decode_frames(d_bits, 2, d_out, d_res, tie_rule=tie)
        torch.cuda.synchronize()
        out, res = d_out.cpu().numpy(), results_np(d_res)
        eb, em, ee = oracle.fic_decode_group(zeros, tie)
        assert np.array_equal(out[0, 0], eb) and int(res[0]["path_error"]) == ee
        eb, em, ee = oracle.fic_decode_group(np.maximum(sat, -127), tie)
        assert np.array_equal(out[1, 0], eb) and int(res[4]["crc_ok_mask"]) == 7 and int(res[4]["path_error"]) == ee
    bad = dabgpu.Codeword()
    bad.d_src = bad.d_out = d_bits.data_ptr()
    bad.n_steps = 100
    with pytest.raises(dabgpu.DabGpuError):
        ctx.viterbi_decode_batch([bad], d_res)


@pytest.mark.parametrize("layout", [0, 1], ids=["natural", "classed"])
@pytest.mark.parametrize("tie_rule", [0, 1])
def test_msc_frames_with_history_ring(ctx, oracle, tie_rule, mapping, layout):
    """3 ensembles x mixed multiplex (EEP-A, EEP-B, 2-A special, UEP) through 7 frames (28 CIFs): the kernel's
    de-interleave-by-index over the frame-history ring must equal CIF_Deinterleaver + MSC_Decoder of the oracle --
    with the history in On_OFDM_Frame() order and in time-interleaver class order (DABGPU_BITS_MSC_CLASSED)"""
    import dabgpu
    import torch
    to_classed = np.argsort(dabgpu.classed_to_natural_index())      # classed_frame = natural_frame[to_classed]
    rng = np.random.default_rng(31 + tie_rule)
    subs = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(48, 8, eep_level=1, eep_type=0),
            oracle.subchannel(60, 27, eep_level=0, eep_type=1), oracle.subchannel(100, 35, is_uep=True, uep_index=4),
            oracle.subchannel(700, 164, eep_level=3, eep_type=0)]
    gsubs = [dabgpu.SubChannel(s.start_address, s.length, s.is_uep, s.uep_prot_index, s.eep_prot_level, s.eep_type) for s in subs]
    plans = [oracle.subchannel_plan(s) for s in subs]
    for s, g in zip(subs, gsubs):                                    # product's host tables == oracle's
        pi, lx, nb = dabgpu.subchannel_plan(g)
        opi, olx, onb = oracle.subchannel_plan(s)
        assert pi == [int(v) for v in opi] and lx == [int(v) for v in olx] and nb == onb
    n_ens, n_frames, H = 3, 7, 5
    n_cif = 4 * n_frames
    cif_out = sum(p[2] for p in plans)
    cifs = rng.integers(-127, 128, (n_ens, n_cif, oracle.NB_CIF_BITS), dtype=np.int8)        # other CUs: junk
    for e in range(n_ens):
        for s, p in zip(subs, plans):
            lf = np.stack([oracle.msc_encode_logical(s, rng.integers(0, 256, p[2], dtype=np.uint8)) for _ in range(n_cif)])
            tx = oracle.time_interleave(lf)
            for t in range(n_cif):
                cifs[e, t, s.start_address * 64:(s.start_address + s.length) * 64] = noisy(oracle, tx[t], rng, gain=0.5, sigma=24.0)
    hist = torch.zeros((n_ens, H, oracle.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
    deint = [[oracle.Deinterleaver(s.length * 8) for s in subs] for _ in range(n_ens)]
    checked = 0
    for f in range(n_frames):
        slot = f % H
        frame = np.zeros((n_ens, oracle.NB_FRAME_BITS), np.int8)
        frame[:, 9216:] = cifs[:, 4 * f:4 * f + 4].reshape(n_ens, -1)
        if layout:
            frame = np.ascontiguousarray(frame[:, to_classed])
        hist[:, slot].copy_(torch.from_numpy(frame).cuda())
        ctx.msc_decode_frames(hist, n_ens, H * oracle.NB_FRAME_BITS, H, slot, gsubs, d_out, 4 * cif_out, d_res, tie_rule=tie_rule,
                              bits_layout=layout)
        torch.cuda.synchronize()
        out, res = d_out.cpu().numpy(), results_np(d_res).reshape(n_ens, 4, len(subs))
        for e in range(n_ens):
            for c in range(4):
                off = 0
                for si, (s, p) in enumerate(zip(subs, plans)):
                    deint[e][si].consume(cifs[e, 4 * f + c, s.start_address * 64:(s.start_address + s.length) * 64])
                    lfr = deint[e][si].deinterleave()
                    if lfr is not None:
                        eb, ee = oracle.msc_decode_logical(s, lfr, tie_rule)
                        assert np.array_equal(out[e, c, off:off + p[2]], eb), (f, e, c, si)
                        assert int(res[e, c, si]["path_error"]) == ee and int(res[e, c, si]["n_out_bytes"]) == p[2]
                        checked += 1
                    off += p[2]
    assert checked == n_ens * len(subs) * (n_cif - 15)


@pytest.fixture(params=[0, 2], ids=["one_launch", "sliced"])
def scratch_mb(request):
    """DABGPU_VIT_SCRATCH_MB = 2 forces the lane mapping to run the batch as several launches over slices of it"""
    import os
    if request.param:
        os.environ["DABGPU_VIT_SCRATCH_MB"] = str(request.param)
    yield request.param
    os.environ.pop("DABGPU_VIT_SCRATCH_MB", None)


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_lane_mapping_equals_wave_mapping_on_a_batch(ctx, oracle, tie_rule, scratch_mb):
    """a batch large enough for several full and one ragged group per schedule: FIC of 333 frames (1332 codewords) and an
    MSC multiplex of seven sub-channels (DAB sub-channel sizes always give n_steps mod 6 = 0; the other start phases of the
    rotating layout are covered by test_uniform_generic_batch_through_both_mappings), with skipped
    ensembles (ring decode), clean / noisy / saturated / all-erased inputs -- bytes, CRC masks and path errors identical"""
    import dabgpu
    import torch
    rng = np.random.default_rng(77 + tie_rule)
    n_frames = 333
    frames = rng.integers(-127, 128, (n_frames, oracle.NB_FRAME_BITS), dtype=np.int8)
    for f in range(0, n_frames, 3):
        for g in range(4):
            enc = oracle.fic_encode_group(rng.integers(0, 256, 90, dtype=np.uint8))
            frames[f, g * 2304:(g + 1) * 2304] = noisy(oracle, enc, rng, sigma=[0.0, 25.0, 45.0][(f // 3 + g) % 3] + 1e-3)
    frames[1, :2304] = 0
    frames[4, :2304] = -128
    frames[7, :9216] = 127
    d_bits = torch.from_numpy(frames).cuda()
    got = {}
    for m in (1, 2, 3):
        ctx.viterbi_set_mapping(m)
        d_out = torch.zeros((n_frames, 4, 96), dtype=torch.uint8, device="cuda")
        d_res = torch.zeros((n_frames * 4, 16), dtype=torch.uint8, device="cuda")
        ctx.fic_decode_frames(d_bits, n_frames, d_out, d_res, tie_rule=tie_rule)
        torch.cuda.synchronize()
        got[m] = (d_out.cpu().numpy(), d_res.cpu().numpy())
    ctx.viterbi_set_mapping(0)
    assert np.array_equal(got[1][0], got[2][0]) and np.array_equal(got[1][1], got[2][1])
    assert np.array_equal(got[1][0], got[3][0]) and np.array_equal(got[1][1], got[3][1])
    assert 0 < int((results_np(torch.from_numpy(got[2][1]))["crc_ok_mask"] == 7).sum()) < n_frames * 4

    subs = [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(48, 8, False, 0, 1, 0), dabgpu.SubChannel(60, 27, False, 0, 0, 1),
            dabgpu.SubChannel(100, 35, True, 4, 0, 0), dabgpu.SubChannel(700, 164, False, 0, 3, 0), dabgpu.SubChannel(300, 4, False, 0, 3, 0), dabgpu.SubChannel(310, 8, False, 0, 3, 0)]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 37, 6
    hist = torch.from_numpy(rng.integers(-127, 128, (n_ens, H, oracle.NB_FRAME_BITS), dtype=np.int8)).cuda()
    hist[3] = 0
    hist[5] = 127
    slots = torch.from_numpy(rng.integers(-1, H, n_ens).astype(np.int32)).cuda()
    got = {}
    import os
    for m in (1, 2, 3, 0):                                # 0 = AUTO with a forced hybrid: the 3 longest sub-channels by WAVE, 4 by LANE / OCTET
        ctx.viterbi_set_mapping(m)
        if m == 0:
            os.environ["DABGPU_VIT_HYBRID_K"] = "3"
        for ring in (False, True):
            d_out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            if ring:
                ctx.msc_decode_ring(hist, n_ens, H * oracle.NB_FRAME_BITS, H, slots, subs, d_out, 4 * cif_out, d_res, tie_rule=tie_rule)
            else:
                ctx.msc_decode_frames(hist, n_ens, H * oracle.NB_FRAME_BITS, H, 2, subs, d_out, 4 * cif_out, d_res, tie_rule=tie_rule)
            torch.cuda.synchronize()
            got[(m, ring)] = (d_out.cpu().numpy(), d_res.cpu().numpy())
    ctx.viterbi_set_mapping(0)
    os.environ.pop("DABGPU_VIT_HYBRID_K", None)
    for ring in (False, True):
        for m in (2, 3, 0):
            assert np.array_equal(got[(1, ring)][0], got[(m, ring)][0]), (m, ring)
            assert np.array_equal(got[(1, ring)][1], got[(m, ring)][1]), (m, ring)


@pytest.mark.parametrize("tie_rule", [0, 1])
@pytest.mark.parametrize("segs", [((3, 40), (0, 0), (24, 24), (11, 72)), ((17, 80),), ((8, 8),), ((1, 2048),)],
                         ids=["T142", "T86", "T14", "T2054"])
def test_uniform_generic_batch_through_both_mappings(ctx, oracle, tie_rule, segs):
    """70 codewords of ONE schedule (segments in 8-step units, an empty segment, n_steps mod 6 = 4 / 2 / 2 / 2), random
    soft bits (every decision is a near tie), random start / end states, raw and descrambled outputs: oracle == WAVE == LANE == OCTET"""
    import dabgpu
    import torch
    rng = np.random.default_rng(500 + tie_rule + len(segs))
    n_cw = 70
    n_bits = sum(st for _, st in segs)
    n_in = sum(st // 8 * (8 + pi) for pi, st in segs) + 12
    soft = rng.integers(-128, 128, (n_cw, n_in), dtype=np.int8)
    soft[3] = 0
    soft[4, ::2] = 127
    starts, ends = rng.integers(0, 64, n_cw), rng.integers(0, 64, n_cw)
    raws = rng.integers(0, 2, n_cw)
    d_in = torch.from_numpy(soft).cuda()
    expect = []
    for i in range(n_cw):
        v = oracle.Viterbi(n_bits, tie_rule)
        v.reset(int(starts[i]))
        x = np.maximum(soft[i], -127)
        used = 0
        for pi, st in segs:
            if st:
                used += v.update(x[used:], oracle.puncture_code(pi), 4 * st)
        used += v.update(x[used:], oracle.puncture_code_tail(), 24)
        assert used == n_in
        ob, oe = v.chainback(n_bits // 8, int(ends[i]))
        if not raws[i]:
            ob = ob ^ oracle.scrambler_bytes(n_bits // 8)
        expect.append((ob, oe))
    for m in (1, 2, 3):
        ctx.viterbi_set_mapping(m)
        d_o = torch.zeros((n_cw, n_bits // 8), dtype=torch.uint8, device="cuda")
        d_res = torch.zeros((n_cw, 16), dtype=torch.uint8, device="cuda")
        cws = []
        for i in range(n_cw):
            cw = dabgpu.Codeword()
            cw.d_src, cw.d_out, cw.n_steps = d_in[i].data_ptr(), d_o[i].data_ptr(), n_bits + 6
            for k, (pi, st) in enumerate(segs):
                cw.seg_pi[k], cw.seg_steps[k] = pi, st
            cw.start_state, cw.end_state, cw.flags = int(starts[i]), int(ends[i]), int(raws[i])
            cws.append(cw)
        ctx.viterbi_decode_batch(cws, d_res, tie_rule=tie_rule)
        torch.cuda.synchronize()
        out, res = d_o.cpu().numpy(), results_np(d_res)
        for i, (ob, oe) in enumerate(expect):
            assert np.array_equal(out[i], ob), (m, i)
            assert int(res[i]["path_error"]) == oe and int(res[i]["n_out_bytes"]) == n_bits // 8, (m, i)
    ctx.viterbi_set_mapping(0)


@pytest.mark.parametrize("warm", ["0", "1"])
def test_octet_mapping_when_the_split_chain_back_fails_its_check(ctx, oracle, warm):
    """DABGPU_VIT_MAP_OCTET splits the chain-back of a codeword over its 8 lanes: each lane starts early from an arbitrary state and the
    positions are checked link by link; a lane whose assumption was wrong walks again from the true position, round after round.
    Without a run-in (DABGPU_VIT_OCTET_WARM=0) nearly every link fails, with 24 steps of it a few do: the bytes, CRC masks and path
    errors must not depend on how many rounds it took.  FIB groups against the oracle, a long and a short sub-channel against the wave mapping."""
    import os
    import dabgpu
    import torch
    rng = np.random.default_rng(31)
    n_frames = 21
    frames = np.zeros((n_frames, oracle.NB_FRAME_BITS), dtype=np.int8)
    for f in range(n_frames):
        for g in range(4):
            enc = oracle.fic_encode_group(rng.integers(0, 256, 90, dtype=np.uint8))
            frames[f, g * 2304:(g + 1) * 2304] = noisy(oracle, enc, rng, sigma=[0.0, 25.0, 45.0, 80.0][(f + g) % 4] + 1e-3)
    frames[3, :2304] = 0
    frames[4, 2304:4608] = rng.integers(-127, 128, 2304, dtype=np.int8)      # pure noise: survivors merge late
    d_bits = torch.from_numpy(frames).cuda()
    subs = [dabgpu.SubChannel(0, 164, False, 0, 3, 0), dabgpu.SubChannel(300, 4, False, 0, 3, 0), dabgpu.SubChannel(400, 35, True, 4, 0, 0)]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 11, 5
    hist = torch.from_numpy(rng.integers(-127, 128, (n_ens, H, oracle.NB_FRAME_BITS), dtype=np.int8)).cuda()
    got = {}
    os.environ["DABGPU_VIT_OCTET_WARM"] = warm
    try:
        for m in (1, 3):
            ctx.viterbi_set_mapping(m)
            d_out = torch.zeros((n_frames, 4, 96), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_frames * 4, 16), dtype=torch.uint8, device="cuda")
            ctx.fic_decode_frames(d_bits, n_frames, d_out, d_res, tie_rule=0)
            m_out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda")
            m_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            ctx.msc_decode_frames(hist, n_ens, H * oracle.NB_FRAME_BITS, H, 1, subs, m_out, 4 * cif_out, m_res)
            torch.cuda.synchronize()
            got[m] = (d_out.cpu().numpy(), d_res.cpu().numpy(), m_out.cpu().numpy(), m_res.cpu().numpy())
    finally:
        os.environ.pop("DABGPU_VIT_OCTET_WARM", None)
        ctx.viterbi_set_mapping(0)
    for k in range(4):
        assert np.array_equal(got[1][k], got[3][k]), k
    out, res = got[3][0], results_np(torch.from_numpy(got[3][1]))
    for f in range(n_frames):
        for g in range(4):
            eb, em, ee = oracle.fic_decode_group(frames[f, g * 2304:(g + 1) * 2304], 0)
            assert np.array_equal(out[f, g], eb) and int(res[f * 4 + g]["crc_ok_mask"]) == em and int(res[f * 4 + g]["path_error"]) == ee
