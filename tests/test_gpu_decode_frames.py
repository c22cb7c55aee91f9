"""-m gpu: dabgpu_decode_frames_layout / dabgpu_decode_ring_layout (FIC + MSC of a transmission frame in one call, include/dabgpu.h).
The one-call form must return exactly what dabgpu_fic_decode_frames + dabgpu_msc_decode_frames_layout return -- FIB bytes, CRC masks,
sub-channel bytes, every result record -- whichever way it runs: the FIB groups inside the MSC launch (lane / octet mapping, one slice)
or the two decodes one after the other (wave mapping, hybrids, sliced batches); and what it returns is what the oracle decodes
(oracle/: fic_decode_group, Deinterleaver + msc_decode_logical)."""
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


def _subs(dabgpu):
    return [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(48, 8, False, 0, 1, 0), dabgpu.SubChannel(60, 27, False, 0, 0, 1),
            dabgpu.SubChannel(100, 35, True, 4, 0, 0), dabgpu.SubChannel(700, 164, False, 0, 3, 0), dabgpu.SubChannel(400, 96, False, 0, 1, 0)]


def _history(rng, n_ens, H, with_fibs):
    """random soft bits; with_fibs: the FIC of every frame carries valid FIBs (so that CRC masks are not all zero)"""
    nat = rng.integers(-127, 128, (n_ens, H, 230400), dtype=np.int8)
    if n_ens > 2:
        nat[1] = 0
        nat[2] = 127
    return nat


@pytest.fixture(params=[0, 2], ids=["one_launch", "sliced"])
def scratch_mb(request):
    if request.param:
        os.environ["DABGPU_VIT_SCRATCH_MB"] = str(request.param)
    yield request.param
    os.environ.pop("DABGPU_VIT_SCRATCH_MB", None)


@pytest.mark.parametrize("tie_rule", [0, 1])
@pytest.mark.parametrize("n_ens", [3, 37, 130])
@pytest.mark.parametrize("layout", [0, 1], ids=["natural", "classed"])
def test_one_call_equals_the_two_calls(ctx, tie_rule, n_ens, layout, scratch_mb):
    import dabgpu
    import torch
    rng = np.random.default_rng(7000 + 10 * n_ens + tie_rule + 3 * layout)
    subs = _subs(dabgpu)
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    H = 6
    nat = _history(rng, n_ens, H, False)
    if layout:
        nat = np.ascontiguousarray(nat[:, :, np.argsort(dabgpu.classed_to_natural_index())])
    hist = torch.from_numpy(nat).cuda()
    for mapping in (0, 1, 2, 3):                              # AUTO, WAVE (two calls), LANE, OCTET (FIB groups inside the MSC launch)
        for hybrid in ((None, "2") if mapping == 0 else (None,)):
            ctx.viterbi_set_mapping(mapping)
            if hybrid:
                os.environ["DABGPU_VIT_HYBRID_K"] = hybrid
            try:
                for slot in (0, 4):
                    fib_a = torch.zeros((n_ens, 4, 96), dtype=torch.uint8, device="cuda"); fres_a = torch.zeros((n_ens * 4, 16), dtype=torch.uint8, device="cuda")
                    out_a = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda"); res_a = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
                    fib_b, fres_b, out_b, res_b = (torch.full_like(x, 0xEE) for x in (fib_a, fres_a, out_a, res_a))
                    ctx.fic_decode_frames(hist[:, slot], n_ens, fib_a, fres_a, frame_stride=H * 230400, tie_rule=tie_rule)
                    ctx.msc_decode_frames(hist, n_ens, H * 230400, H, slot, subs, out_a, 4 * cif_out, res_a, tie_rule=tie_rule, bits_layout=layout)
                    ctx.decode_frames(hist, n_ens, H * 230400, H, slot, subs, fib_b, fres_b, out_b, 4 * cif_out, res_b, tie_rule=tie_rule, bits_layout=layout)
                    torch.cuda.synchronize()
                    tag = (mapping, hybrid, slot)
                    assert torch.equal(fib_a, fib_b), tag
                    assert torch.equal(fres_a, fres_b), tag
                    assert torch.equal(out_a, out_b), tag
                    assert torch.equal(res_a, res_b), tag
                    assert out_a.any() and fib_a.any()
            finally:
                os.environ.pop("DABGPU_VIT_HYBRID_K", None)
    ctx.viterbi_set_mapping(0)


def test_ring_form_with_skipped_ensembles(ctx):
    """every ensemble at its own ring slot, some skipped (slot -1): one call == dabgpu_fic_decode_ring + dabgpu_msc_decode_ring_layout"""
    import dabgpu
    import torch
    rng = np.random.default_rng(4242)
    subs = _subs(dabgpu)
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 70, 8
    nat = _history(rng, n_ens, H, False)
    for layout in (0, 1):
        h = np.ascontiguousarray(nat[:, :, np.argsort(dabgpu.classed_to_natural_index())]) if layout else nat
        hist = torch.from_numpy(h).cuda()
        slots = rng.integers(0, H, n_ens).astype(np.int32)
        slots[[0, 5, 17, 64, 69]] = -1
        d_slots = torch.from_numpy(slots).cuda()
        for mapping in (1, 2, 3):
            ctx.viterbi_set_mapping(mapping)
            fib_a = torch.zeros((n_ens, 4, 96), dtype=torch.uint8, device="cuda"); fres_a = torch.zeros((n_ens * 4, 16), dtype=torch.uint8, device="cuda")
            out_a = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda"); res_a = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            fib_b, fres_b, out_b, res_b = (torch.zeros_like(x) for x in (fib_a, fres_a, out_a, res_a))
            ctx.fic_decode_ring(hist, n_ens, H * 230400, d_slots, fib_a, fres_a)
            ctx.msc_decode_ring(hist, n_ens, H * 230400, H, d_slots, subs, out_a, 4 * cif_out, res_a, bits_layout=layout)
            ctx.decode_ring(hist, n_ens, H * 230400, H, d_slots, subs, fib_b, fres_b, out_b, 4 * cif_out, res_b, bits_layout=layout)
            torch.cuda.synchronize()
            assert torch.equal(fib_a, fib_b) and torch.equal(fres_a, fres_b) and torch.equal(out_a, out_b) and torch.equal(res_a, res_b), (layout, mapping)
            assert not out_b[0].any() and not fib_b[5].any() and out_b[1].any()
    ctx.viterbi_set_mapping(0)


def test_one_call_against_the_oracle(ctx, oracle):
    """5 ensembles, lane mapping (FIB groups inside the MSC launch): every FIB byte, CRC mask, path error and sub-channel byte is the oracle's"""
    import dabgpu
    import torch
    rng = np.random.default_rng(99)
    subs_o = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(120, 27, eep_level=0, eep_type=1)]
    subs = [dabgpu.SubChannel(0, 48, False, 0, 2, 0), dabgpu.SubChannel(120, 27, False, 0, 0, 1)]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    n_ens, H = 5, 5
    nat = rng.integers(-127, 128, (n_ens, H, 230400), dtype=np.int8)
    hist = torch.from_numpy(nat).cuda()
    newest = 4
    ctx.viterbi_set_mapping(2)
    try:
        fib = torch.zeros((n_ens, 4, 96), dtype=torch.uint8, device="cuda"); fres = torch.zeros((n_ens * 4, 16), dtype=torch.uint8, device="cuda")
        out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda"); res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
        ctx.decode_frames(hist, n_ens, H * 230400, H, newest, subs, fib, fres, out, 4 * cif_out, res)
        torch.cuda.synchronize()
    finally:
        ctx.viterbi_set_mapping(0)
    fib, out = fib.cpu().numpy(), out.cpu().numpy()
    fres = fres.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(n_ens, 4)
    res = res.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(n_ens, 4, len(subs))
    for e in range(n_ens):
        for g in range(4):
            eb, em, ee = oracle.fic_decode_group(nat[e, newest, g * 2304:(g + 1) * 2304], 0)
            assert np.array_equal(fib[e, g], eb) and int(fres["crc_ok_mask"][e, g]) == em and int(fres["path_error"][e, g]) == ee
        for si, s in enumerate(subs_o):
            d = oracle.Deinterleaver(s.length * 8)
            outs = []
            for f in range(H):                                   # frames 0..4 in ring order (slot = frame here), 4 CIFs each
                for c in range(4):
                    cif = nat[e, f, 9216 + c * 55296: 9216 + (c + 1) * 55296]
                    d.consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                    outs.append(d.deinterleave())
            off = sum(dabgpu.subchannel_plan(g)[2] for g in subs[:si])
            for c in range(4):
                lf = outs[4 * newest + c]
                assert lf is not None
                dec, err = oracle.msc_decode_logical(s, lf, 0)
                assert np.array_equal(out[e, c, off:off + dec.size], dec), (e, si, c)
                assert int(res["path_error"][e, c, si]) == err


def test_degenerate_arguments(ctx):
    """no ensembles: nothing happens; no sub-channels: the FIC alone; a slot outside the ring, a missing FIC array: refused"""
    import dabgpu
    import torch
    rng = np.random.default_rng(1)
    n_ens, H = 9, 5
    hist = torch.from_numpy(rng.integers(-127, 128, (n_ens, H, 230400), dtype=np.int8)).cuda()
    subs = _subs(dabgpu)[:2]
    cif_out = sum(dabgpu.subchannel_plan(g)[2] for g in subs)
    fib = torch.zeros((n_ens, 4, 96), dtype=torch.uint8, device="cuda"); fres = torch.zeros((n_ens * 4, 16), dtype=torch.uint8, device="cuda")
    out = torch.zeros((n_ens, 4, cif_out), dtype=torch.uint8, device="cuda"); res = torch.zeros((n_ens * 4 * len(subs), 16), dtype=torch.uint8, device="cuda")
    ctx.decode_frames(hist, 0, H * 230400, H, 4, subs, fib, fres, out, 4 * cif_out, res)
    torch.cuda.synchronize()
    assert not fib.any() and not out.any()
    ctx.decode_frames(hist, n_ens, H * 230400, H, 4, [], fib, fres, None, 0, None)            # FIC only
    ref_f = torch.zeros_like(fib); ref_r = torch.zeros_like(fres)
    ctx.fic_decode_frames(hist[:, 4], n_ens, ref_f, ref_r, frame_stride=H * 230400)
    torch.cuda.synchronize()
    assert torch.equal(fib, ref_f) and torch.equal(fres, ref_r) and fib.any()
    with pytest.raises(RuntimeError):
        ctx.decode_frames(hist, n_ens, H * 230400, H, H, subs, fib, fres, out, 4 * cif_out, res)       # newest slot outside the ring
    with pytest.raises(RuntimeError):
        ctx.decode_frames(hist, n_ens, H * 230400, H, 4, subs, None, fres, out, 4 * cif_out, res)       # no FIB array
    with pytest.raises(RuntimeError):
        ctx.decode_frames(hist, n_ens, H * 230400, H, 4, subs, fib, fres, out, cif_out, res)            # output stride too small
