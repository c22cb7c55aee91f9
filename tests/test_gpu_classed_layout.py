"""-m gpu: the time-interleaver class order of the MSC soft bits (DABGPU_BITS_MSC_CLASSED, include/dabgpu.h) is a pure re-arrangement:
the demodulator's class-order output is the natural output permuted (every capture format with a fused loader), and the channel decoder
gives identical bytes, CRC-less results and path errors from either order, for both device mappings, a forced hybrid, both tie rules,
ragged ensemble counts and batches large enough for several gather workgroups per group.  (Parity with the oracle of the class-order
path itself: tests/test_gpu_viterbi.py::test_msc_frames_with_history_ring[classed-*].)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


def test_index_helper_is_the_documented_permutation():
    import dabgpu
    P = dabgpu.classed_to_natural_index()
    assert P.shape == (230400,) and np.array_equal(np.sort(P), np.arange(230400)) and np.array_equal(P[:9216], np.arange(9216))
    for q in (0, 3):
        for i in (0, 1, 15, 16, 17, 3455 * 16 + 15, 55295):
            assert P[9216 + q * 55296 + i] == 9216 + q * 55296 + (i % 16) * 3456 + i // 16


@pytest.mark.parametrize("fmt_name", ["raw_f32l", "raw_u8", "raw_s8", "raw_s16l"])
@pytest.mark.parametrize("stride_extra", [0, 4 * 230400])
def test_demodulator_class_order_is_the_natural_output_permuted(ctx, fmt_name, stride_extra):
    import dabgpu
    import torch
    rng = np.random.default_rng(5)
    n = 7
    fmt = dabgpu.IQ_FORMATS.index(fmt_name)
    if fmt_name == "raw_f32l":
        raw = torch.from_numpy(rng.standard_normal((n, 196608, 2)).astype(np.float32)).cuda()
    elif fmt_name == "raw_s16l":
        raw = torch.from_numpy(rng.integers(-30000, 30000, (n, 196608, 2), dtype=np.int16)).cuda()
    elif fmt_name == "raw_s8":
        raw = torch.from_numpy(rng.integers(-128, 128, (n, 196608, 2), dtype=np.int8)).cuda()
    else:
        raw = torch.from_numpy(rng.integers(0, 256, (n, 196608, 2), dtype=np.uint8)).cuda()
    freq = torch.from_numpy(((rng.random(n) * 2 - 1) * 2.0e-3).astype(np.float32)).cuda()
    stride = 230400 + stride_extra
    outs, corrs = [], []
    for layout in (dabgpu.BITS_NATURAL, dabgpu.BITS_MSC_CLASSED):
        for spb in (0, 7, 75):
            bits = torch.full((n, stride), 99, dtype=torch.int8, device="cuda")
            corr = torch.zeros((n, 76, 2), dtype=torch.float32, device="cuda")
            ctx.ofdm_demod_frames_history(raw, fmt, n, bits, freq_offset=freq, cp_corr=corr, symbols_per_block=spb,
                                          bits_frame_stride=stride if stride_extra else 0, bits_layout=layout)
            torch.cuda.synchronize()
            b = bits.cpu().numpy()
            assert stride_extra == 0 or (b[:, 230400:] == 99).all(), "nothing is written past a frame's 230400 soft bits"
            outs.append(b[:, :230400]); corrs.append(corr.cpu().numpy())
    ref = torch.zeros((n, 230400), dtype=torch.int8, device="cuda")
    ctx.ofdm_demod_frames_raw(raw, fmt, n, ref, freq_offset=freq)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    assert len(np.unique(ref)) > 100
    P = dabgpu.classed_to_natural_index()
    for k in range(3):
        assert np.array_equal(outs[k], ref), "natural layout == dabgpu_ofdm_demod_frames_raw"
        assert np.array_equal(outs[3 + k][:, P], ref), "class order == the natural output permuted"
        assert np.array_equal(corrs[k].view(np.uint32), corrs[0].view(np.uint32)) and np.array_equal(corrs[3 + k].view(np.uint32), corrs[0].view(np.uint32))


@pytest.fixture(params=[0, 2], ids=["one_launch", "sliced"])
def scratch_mb(request):
    """DABGPU_VIT_SCRATCH_MB = 2 forces the lane mapping to run the batch as several launches over slices of the ensembles"""
    if request.param:
        os.environ["DABGPU_VIT_SCRATCH_MB"] = str(request.param)
    yield request.param
    os.environ.pop("DABGPU_VIT_SCRATCH_MB", None)


@pytest.mark.parametrize("tie_rule", [0, 1])
@pytest.mark.parametrize("n_ens", [5, 37, 130])
def test_decoder_gives_identical_results_from_either_order(ctx, tie_rule, n_ens, scratch_mb):
    import dabgpu
    import torch
    rng = np.random.default_rng(100 + n_ens + tie_rule)
    subs = [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(48, 8, False, 0, 1, 0), dabgpu.SubChannel(60, 27, False, 0, 0, 1),
            dabgpu.SubChannel(100, 35, True, 4, 0, 0), dabgpu.SubChannel(700, 164, False, 0, 3, 0), dabgpu.SubChannel(300, 4, False, 0, 3, 0),
            dabgpu.SubChannel(310, 8, False, 0, 3, 0), dabgpu.SubChannel(400, 96, False, 0, 1, 0)]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    H = 6
    nat = rng.integers(-128, 128, (n_ens, H, 230400), dtype=np.int8)
    nat[1] = 0
    nat[2] = 127
    to_classed = np.argsort(dabgpu.classed_to_natural_index())
    hists = {0: torch.from_numpy(nat).cuda(), 1: torch.from_numpy(np.ascontiguousarray(nat[:, :, to_classed])).cuda()}
    got = {}
    for m in (1, 2, 3, 0):                                # WAVE, LANE, OCTET, AUTO with a forced hybrid (3 longest sub-channels by WAVE)
        ctx.viterbi_set_mapping(m)
        if m == 0:
            os.environ["DABGPU_VIT_HYBRID_K"] = "3"
        try:
            for layout in (0, 1):
                for slot in (0, 4):
                    d_out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda")
                    d_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
                    ctx.msc_decode_frames(hists[layout], n_ens, H * 230400, H, slot, subs, d_out, 4 * cif_out, d_res, tie_rule=tie_rule,
                                          bits_layout=layout)
                    torch.cuda.synchronize()
                    got[(m, layout, slot)] = (d_out.cpu().numpy(), d_res.cpu().numpy())
        finally:
            os.environ.pop("DABGPU_VIT_HYBRID_K", None)
    ctx.viterbi_set_mapping(0)
    for slot in (0, 4):
        base = got[(1, 0, slot)]
        assert base[0].any()
        for m in (1, 2, 3, 0):
            for layout in (0, 1):
                assert np.array_equal(got[(m, layout, slot)][0], base[0]), (m, layout, slot)
                assert np.array_equal(got[(m, layout, slot)][1], base[1]), (m, layout, slot)


@pytest.mark.parametrize("level,type_b", [(3, 0), (0, 0), (3, 1)])
def test_one_sub_channel_filling_the_cif(ctx, level, type_b):
    """864 CU in one sub-channel (up to 44,000 trellis steps: more than a hundred gather tiles per group, the last memory line of the
    last class segment of every row in use): class order == natural order through the lane mapping, == the wave mapping"""
    import dabgpu
    import torch
    rng = np.random.default_rng(9 + level)
    subs = [dabgpu.SubChannel(0, 864, False, 0, level, type_b)]
    nb = dabgpu.subchannel_plan(subs[0])[2]
    n_ens, H = 3, 5
    nat = rng.integers(-128, 128, (n_ens, H, 230400), dtype=np.int8)
    to_classed = np.argsort(dabgpu.classed_to_natural_index())
    hists = {0: torch.from_numpy(nat).cuda(), 1: torch.from_numpy(np.ascontiguousarray(nat[:, :, to_classed])).cuda()}
    got = {}
    for m in (1, 2, 3):
        ctx.viterbi_set_mapping(m)
        for layout in (0, 1):
            d_out = torch.zeros((n_ens, 4, nb), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_ens * 4, 16), dtype=torch.uint8, device="cuda")
            ctx.msc_decode_frames(hists[layout], n_ens, H * 230400, H, 3, subs, d_out, 4 * nb, d_res, bits_layout=layout)
            torch.cuda.synchronize()
            got[(m, layout)] = (d_out.cpu().numpy(), d_res.cpu().numpy())
    ctx.viterbi_set_mapping(0)
    for key in ((1, 1), (2, 0), (2, 1), (3, 0), (3, 1)):
        assert np.array_equal(got[key][0], got[(1, 0)][0]) and np.array_equal(got[key][1], got[(1, 0)][1]), key


@pytest.mark.parametrize("mapping", [1, 2, 3], ids=["wave", "lane", "octet"])
def test_generic_batch_with_a_classed_ring_of_any_geometry(ctx, mapping):
    """DABGPU_CW_CLASSED through dabgpu_viterbi_decode_batch: 70 codewords of one schedule, each with its own 16-slot ring of one
    sub-channel per row (cifs_per_frame = 1, cif_stride = the sub-channel's soft bits) -- the general readers (wave-mapped decoder,
    byte gather of the lane mapping) must give the same bytes and path errors from the class-ordered rows as from the natural ones"""
    import dabgpu
    import torch
    rng = np.random.default_rng(21)
    n_cw = 70                                             # EEP 3-A on 24 CU: the (PI, L) plan comes from the library
    sub = dabgpu.SubChannel(0, 24, False, 0, 2, 0)
    pi, lx, nb = dabgpu.subchannel_plan(sub)
    n_in = 24 * 64
    rows = rng.integers(-128, 128, (n_cw, 16, n_in), dtype=np.int8)
    i = np.arange(n_in)
    perm = np.empty(n_in, np.int64)
    perm[(i % 16) * (n_in // 16) + i // 16] = i            # classed_row = natural_row[perm]
    outs = {}
    ctx.viterbi_set_mapping(mapping)
    try:
        for layout in (0, 1):
            d_ring = torch.from_numpy(np.ascontiguousarray(rows[:, :, perm]) if layout else rows).cuda()
            d_out = torch.zeros((n_cw, nb), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_cw, 16), dtype=torch.uint8, device="cuda")
            cws = []
            for k in range(n_cw):
                cw = dabgpu.Codeword()
                cw.d_src, cw.d_out = d_ring[k].data_ptr(), d_out[k].data_ptr()
                cw.n_steps = 32 * sum(lx) + 6
                for q in range(len(lx)):
                    cw.seg_pi[q], cw.seg_steps[q] = (pi[q] if lx[q] else 0), 32 * lx[q]
                cw.n_slots, cw.newest_slot, cw.cifs_per_frame, cw.frame_stride, cw.cif_stride = 16, (k * 5) % 16, 1, n_in, n_in
                cw.flags = 4 if layout else 0                                   # DABGPU_CW_CLASSED
                cws.append(cw)
            ctx.viterbi_decode_batch(cws, d_res, tie_rule=0)
            torch.cuda.synchronize()
            outs[layout] = (d_out.cpu().numpy(), d_res.cpu().numpy())
    finally:
        ctx.viterbi_set_mapping(0)
    assert outs[0][0].any()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    bad = dabgpu.Codeword()
    bad.d_src = bad.d_out = 4096
    bad.n_steps, bad.seg_pi[0], bad.seg_steps[0] = 38, 8, 32
    bad.n_slots, bad.cifs_per_frame, bad.frame_stride, bad.cif_stride, bad.flags = 16, 1, 100, 100, 4
    with pytest.raises(dabgpu.DabGpuError):
        ctx.viterbi_decode_batch([bad], torch.zeros((1, 16), dtype=torch.uint8, device="cuda"))


def test_history_that_is_not_64_byte_aligned_takes_the_general_gather(ctx):
    """the line-streaming gather loads whole 64-byte memory lines and therefore asks for a 64-byte aligned history; a history 16 bytes
    off (and an ensemble stride that is no multiple of 64) must still decode, through the general gather, to the same bytes"""
    import dabgpu
    import torch
    rng = np.random.default_rng(5)
    subs = [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(100, 35, True, 4, 0, 0), dabgpu.SubChannel(700, 164, False, 0, 3, 0)]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 70, 5
    nat = rng.integers(-128, 128, (n_ens, H, 230400), dtype=np.int8)
    to_classed = np.argsort(dabgpu.classed_to_natural_index())
    cls = np.ascontiguousarray(nat[:, :, to_classed])
    stride = H * 230400 + 16                                              # 16 mod 64
    buf = torch.zeros(n_ens * stride + 64, dtype=torch.int8, device="cuda")
    off = (16 - buf.data_ptr() % 64) % 64                                 # data pointer = 16 mod 64
    view = buf[off:off + n_ens * stride].view(n_ens, stride)
    view[:, :H * 230400].copy_(torch.from_numpy(cls.reshape(n_ens, -1)).cuda())
    assert view.data_ptr() % 64 == 16
    aligned = torch.from_numpy(cls).cuda()
    got = []
    ctx.viterbi_set_mapping(2)
    try:
        for hist, st in ((aligned, H * 230400), (view, stride)):
            d_out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            ctx.msc_decode_frames(hist, n_ens, st, H, 2, subs, d_out, 4 * cif_out, d_res, bits_layout=1)
            torch.cuda.synchronize()
            got.append((d_out.cpu().numpy(), d_res.cpu().numpy()))
    finally:
        ctx.viterbi_set_mapping(0)
    assert got[0][0].any() and np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])


def test_bad_layout_and_unsupported_format_are_refused(ctx):
    import dabgpu
    import torch
    raw = torch.zeros((1, 196608, 2), dtype=torch.float32, device="cuda")
    bits = torch.zeros((1, 230400), dtype=torch.int8, device="cuda")
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_demod_frames_history(raw, dabgpu.IQ_FORMATS.index("raw_f32l"), 1, bits, bits_layout=7)
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_demod_frames_history(raw, dabgpu.IQ_FORMATS.index("raw_s16b"), 1, bits, bits_layout=1)


@pytest.mark.parametrize("layout", [0, 1], ids=["natural", "classed"])
def test_ring_decode_with_skipped_ensembles_inside_a_quarter_group(ctx, layout):
    """dabgpu_msc_decode_ring: ensembles without a new frame (slot -1) carry zeroed descriptors.  The class-order gather shares one
    line phase per quarter group of 4 ensembles: it must come from an ensemble that HAS a frame -- here ensemble 0 of most quarters is
    skipped while its neighbours decode, and the last sub-channel ends at CU 864 (the last memory line of the last class segment of a
    row).  The lane mapping must equal the wave mapping, which reads every byte by its own index, and skipped ensembles stay untouched."""
    import dabgpu
    import torch
    rng = np.random.default_rng(77)
    subs = [dabgpu.SubChannel(3, 48, False, 0, 2, 0), dabgpu.SubChannel(804, 60, False, 0, 2, 0)]      # start CUs 3 and 804: line phases 12 and 16
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 37, 6
    nat = rng.integers(-128, 128, (n_ens, H, 230400), dtype=np.int8)
    hist = nat if layout == 0 else np.ascontiguousarray(nat[:, :, np.argsort(dabgpu.classed_to_natural_index())])
    d_hist = torch.from_numpy(hist).cuda()
    slots = rng.integers(0, H, n_ens).astype(np.int32)
    slots[0::4] = -1                                                       # the first ensemble of every quarter group
    slots[5] = -1; slots[6] = -1; slots[7] = -1                            # a quarter with a single active ensemble (4), one with none (8..11 below)
    slots[8:12] = -1
    d_slots = torch.from_numpy(slots).cuda()
    got = {}
    for m in (1, 2, 3):
        ctx.viterbi_set_mapping(m)
        try:
            d_out = torch.full((n_ens, 4, cif_out), 0xEE, dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            ctx.msc_decode_ring(d_hist, n_ens, H * 230400, H, d_slots, subs, d_out, 4 * cif_out, d_res, bits_layout=layout)
            torch.cuda.synchronize()
            got[m] = (d_out.cpu().numpy(), d_res.cpu().numpy())
        finally:
            ctx.viterbi_set_mapping(0)
    assert np.array_equal(got[1][0], got[2][0]) and np.array_equal(got[1][1], got[2][1])
    assert np.array_equal(got[1][0], got[3][0]) and np.array_equal(got[1][1], got[3][1])
    skipped = slots < 0
    assert (got[2][0][skipped] == 0xEE).all(), "no byte is written for an ensemble without a new frame"
    assert (got[2][0][~skipped] != 0xEE).any()
