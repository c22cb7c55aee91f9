"""-m gpu: a HETEROGENEOUS multiplex through the batch decoder (dabgpu_decode_frames_layout), against the oracle.
Every throughput figure of rounds 1-4 was taken on 18 x 48 CU EEP 3-A -- one puncturing schedule, one code word length.  The multiplex
of tools/dabsynth.py::mixed_layout is what is on air: 14 sub-channels of 8 sizes in 4 protection families (EEP 3-A at 48 / 60 / 72 CU,
EEP 2-B, three UEP table rows incl. one with padding bits, the 8 CU EEP 2-A special case; msc_decoder.cpp:77-154,
subchannel_protection_tables.h:21-139).  16 ensembles of it, demodulated from noisy IQ, are decoded by every device mapping (AUTO, WAVE,
LANE, OCTET, a hybrid) in both history layouts; bar: FIB bytes, CRC masks, every sub-channel byte and every result record equal
the oracle's (fic_decode_group, Deinterleaver + msc_decode_logical on the very soft bits the device decoded), whichever mapping ran -- and the
decoded bytes are the transmitted ones."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def received(ctx, oracle):
    """16 ensembles x 6 transmission frames of the mixed multiplex: IQ with noise -> the device's own demodulator -> history rings in both
    layouts; the oracle's decode of the natural-order soft bits of the last frame"""
    import dabgpu
    import dabsynth
    import torch
    dev = torch.device("cuda", 0)
    E, H, n_frames = 16, 8, 6
    prs, mapper, _ = dabgpu.host_tables()
    layout = dabsynth.mixed_layout()
    dabsynth.check_layout(layout, dabgpu)
    iq, mux = dabsynth.ensemble_iq(E, E, 31, dev, mapper, prs, noise=0.09, period=4 * n_frames, layout=layout, fig=True)
    fmt = dabgpu.IQ_FORMATS.index("raw_f32l")
    hist = {lay: torch.zeros((E, H, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev) for lay in (0, 1)}
    for j in range(n_frames):
        for lay in (0, 1):
            ctx.ofdm_demod_frames_history(torch.view_as_real(iq[j]), fmt, E, hist[lay][:, j % H], bits_frame_stride=H * dabgpu.NB_FRAME_BITS, bits_layout=lay)
    torch.cuda.synchronize()
    nat = hist[0].cpu().numpy()
    assert np.array_equal(hist[1].cpu().numpy()[:, :, dabgpu.classed_to_natural_index()], nat), "the class-order history is a permutation of the natural one"
    subs_o = [oracle.subchannel(d["start"], d["length"], eep_level=d["eep_level"], eep_type=d["eep_type"], is_uep=bool(d["is_uep"]), uep_index=d["uep_index"])
              for d in layout]
    # the oracle on the same soft bits: every sub-channel's de-interleaver sees the CIFs of frames 0 .. n_frames - 1 in order
    exp_fib = np.zeros((E, 4, 96), np.uint8); exp_mask = np.zeros((E, 4), np.uint32); exp_ferr = np.zeros((E, 4), np.uint64)
    exp_msc = np.zeros((E, 4, mux.cif_out_bytes), np.uint8); exp_merr = np.zeros((E, 4, len(layout)), np.uint64)
    last = n_frames - 1
    for e in range(E):
        for g in range(4):
            exp_fib[e, g], exp_mask[e, g], exp_ferr[e, g] = oracle.fic_decode_group(nat[e, last % H, g * 2304:(g + 1) * 2304], 0)
        off = 0
        for si, (d, s) in enumerate(zip(layout, subs_o)):
            di = oracle.Deinterleaver(d["length"] * 8)
            for j in range(n_frames):
                for c in range(4):
                    cif = nat[e, j % H, 9216 + c * 55296:9216 + (c + 1) * 55296]
                    di.consume(cif[d["start"] * 64:(d["start"] + d["length"]) * 64])
                    lf = di.deinterleave()
                    if j == last:
                        assert lf is not None
                        dec, err = oracle.msc_decode_logical(s, lf, 0)
                        exp_msc[e, c, off:off + d["nbytes"]] = dec
                        exp_merr[e, c, si] = err
            off += d["nbytes"]
    return dict(E=E, H=H, last=last, mux=mux, layout=layout, hist=hist, exp_fib=exp_fib, exp_mask=exp_mask, exp_ferr=exp_ferr, exp_msc=exp_msc, exp_merr=exp_merr)


@pytest.mark.parametrize("bits_layout", [0, 1], ids=["natural", "classed"])
@pytest.mark.parametrize("mapping,hybrid", [(0, None), (1, None), (2, None), (3, None), (0, "3")], ids=["auto", "wave", "lane", "octet", "hybrid3"])
def test_mixed_multiplex_equals_the_oracle_under_every_mapping(ctx, received, mapping, hybrid, bits_layout):
    import dabgpu
    import torch
    R = received
    E, H, mux = R["E"], R["H"], R["mux"]
    subs = mux.subchannels(dabgpu)
    n_sub, nb = len(subs), mux.cif_out_bytes
    fib = torch.zeros((E, 4, 96), dtype=torch.uint8, device="cuda"); fres = torch.zeros((E * 4, 16), dtype=torch.uint8, device="cuda")
    out = torch.zeros((E, 4, nb), dtype=torch.uint8, device="cuda"); res = torch.zeros((E * 4 * n_sub, 16), dtype=torch.uint8, device="cuda")
    ctx.viterbi_set_mapping(mapping)
    if hybrid:
        os.environ["DABGPU_VIT_HYBRID_K"] = hybrid
    try:
        chosen, model = ctx.multiplex_mapping(E, subs)
        ctx.decode_frames(R["hist"][bits_layout], E, H * dabgpu.NB_FRAME_BITS, H, R["last"] % H, subs, fib, fres, out, 4 * nb, res, bits_layout=bits_layout)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("DABGPU_VIT_HYBRID_K", None)
        ctx.viterbi_set_mapping(0)
    assert chosen == (mapping or chosen) and chosen in (1, 2, 3) and all(v > 0 for v in model.values())
    rdt = np.dtype(dabgpu.RESULT_DTYPE)
    fr = fres.cpu().numpy().view(rdt).reshape(E, 4)
    mr = res.cpu().numpy().view(rdt).reshape(E, 4, n_sub)
    assert np.array_equal(fib.cpu().numpy(), R["exp_fib"])
    assert np.array_equal(fr["crc_ok_mask"], R["exp_mask"]) and np.array_equal(fr["path_error"], R["exp_ferr"])
    got = out.cpu().numpy()
    off = 0
    for si, d in enumerate(R["layout"]):
        assert np.array_equal(got[:, :, off:off + d["nbytes"]], R["exp_msc"][:, :, off:off + d["nbytes"]]), (si, d)
        assert (mr["n_out_bytes"][:, :, si] == d["nbytes"]).all()
        off += d["nbytes"]
    assert np.array_equal(mr["path_error"], R["exp_merr"])
    # ... and the decoded bytes are what was transmitted (CIF r carries the logical frame of CIF r - 15)
    assert (R["exp_mask"] == 7).all()
    for c in range(4):
        assert np.array_equal(got[:, c], mux.expected_cif(4 * R["last"] + c).cpu().numpy())


def test_auto_picks_one_mapping_per_call_from_the_cost_model(ctx):
    """dabgpu_multiplex_mapping reports what the decode call takes: wave for a handful of ensembles, a batch mapping for thousands, and the
    forced setting when there is one"""
    import dabgpu
    import dabsynth
    subs = dabsynth.Multiplex.subchannels(type("L", (), {"layout": dabsynth.mixed_layout()})(), dabgpu)
    small, _ = ctx.multiplex_mapping(4, subs)
    big, model = ctx.multiplex_mapping(4096, subs)
    assert small == 1 and big in (2, 3) and model["lane" if big == 2 else "octet"] < model["wave"]
    ctx.viterbi_set_mapping(3)
    assert ctx.multiplex_mapping(4, subs)[0] == 3
    ctx.viterbi_set_mapping(0)
