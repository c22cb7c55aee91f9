"""-m gpu: the HIP channel decoder against tests/golden/decoder_vectors.npz -- what the REFERENCE's FIC_Decoder / MSC_Decoder / CIF_Deinterleaver code
returned when it was executed (over the restated Viterbi core: tests/golden/make_golden_decoders.py, label in the fixture) on 48 FIB groups and 20 CIFs
of every protection profile of the reference's tables.  Every device mapping (one wavefront / eight lanes / one lane per code word), both core
models, both history layouts where the call has them: FIB bytes + valid count and every decoded byte of CIFs 15..19 of each sub-channel equal the
vectors.  The 20 CIFs sit in an 8-frame history ring as the demodulator would have left them; a call on frame 3 gives the logical frame of CIF 15
(the first one the reference returns, cif_deinterleaver.cpp:40-42), a call on frame 4 those of CIFs 16..19."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def vectors(oracle):
    import decoder_vectors as DV
    import test_decoder_vectors as TD
    fx = TD.load()
    cs, cifs, _ = DV.msc_multiplexes(oracle)
    soft_fic, _ = DV.fic_groups(oracle)
    assert DV.sha(soft_fic) == str(fx["z"]["fic_inputs_sha256"]) and DV.sha(cifs) == str(fx["z"]["msc_inputs_sha256"]), "regenerated inputs differ"
    return dict(fx=fx, cases=cs, cifs=cifs, soft_fic=soft_fic)


@pytest.mark.parametrize("model", [0, 1], ids=["scalar_core", "simd_core"])
@pytest.mark.parametrize("mapping", [1, 2, 3], ids=["wave", "lane", "octet"])
def test_fic_groups_equal_the_reference_decoders_output(ctx, vectors, mapping, model):
    import dabgpu
    import torch
    soft = vectors["soft_fic"]
    n_frames = len(soft) // 4
    frames = np.zeros((n_frames, dabgpu.NB_FRAME_BITS), np.int8)
    frames[:, :9216] = soft.reshape(n_frames, 9216)
    d_bits = torch.from_numpy(frames).cuda()
    d_out = torch.zeros((n_frames, 4, 96), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((n_frames * 4, 16), dtype=torch.uint8, device="cuda")
    ctx.viterbi_set_mapping(mapping)
    try:
        ctx.fic_decode_frames(d_bits, n_frames, d_out, d_res, tie_rule=model)
        torch.cuda.synchronize()
    finally:
        ctx.viterbi_set_mapping(0)
    out = d_out.cpu().numpy().reshape(-1, 96)
    mask = d_res.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(-1)["crc_ok_mask"]
    want_n, want = vectors["fx"]["fic"][model]
    for g in range(len(soft)):
        got = b"".join(out[g, 32 * i:32 * i + 30].tobytes() for i in range(3) if int(mask[g]) & (1 << i))
        assert len(got) == 30 * int(want_n[g]) and got == want[g, :len(got)].tobytes(), g


@pytest.mark.parametrize("model", [0, 1], ids=["scalar_core", "simd_core"])
@pytest.mark.parametrize("bits_layout", [0, 1], ids=["natural", "classed"])
@pytest.mark.parametrize("mapping", [1, 2, 3], ids=["wave", "lane", "octet"])
def test_every_protection_profile_equals_the_reference_decoders_output(ctx, vectors, mapping, bits_layout, model):
    import dabgpu
    import torch
    fx, cs, cifs = vectors["fx"], vectors["cases"], vectors["cifs"]
    H, n_checked, refused = 8, 0, []
    if bits_layout == 1:
        idx = dabgpu.classed_to_natural_index()                      # natural == classed[idx]
    ctx.viterbi_set_mapping(mapping)
    try:
        for m in range(cifs.shape[0]):
            mine = [c for c in cs if c["mux"] == m]
            # UEP row 34 as the reference lists it cannot hold its own code word: the reference decodes what fits (the vectors hold that), the
            # product refuses the descriptor (include/dabgpu.h, dabgpu_host_logic.cpp:331-340) -- asserted below, then left out of the call
            ok, subs = [], []
            for c in mine:
                g = dabgpu.SubChannel(c["start"], c["length"], c["is_uep"], c["uep_index"], c["eep_level"], c["eep_type"])
                pi, lx, nb = dabgpu.subchannel_plan(g)
                if sum(4 * l * (8 + p) for p, l in zip(pi, lx)) + 12 > c["length"] * 64:
                    refused.append(c["name"])
                    d_o = torch.zeros((1, 4, 8), dtype=torch.uint8, device="cuda"); d_r = torch.zeros((4, 16), dtype=torch.uint8, device="cuda")
                    with pytest.raises(dabgpu.DabGpuError):
                        ctx.msc_decode_frames(torch.zeros((1, H, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda"), 1, H * dabgpu.NB_FRAME_BITS, H, 0, [g], d_o, 32, d_r)
                    continue
                ok.append((c, nb)); subs.append(g)
            hist = np.zeros((1, H, dabgpu.NB_FRAME_BITS), np.int8)
            for t in range(cifs.shape[1]):
                hist[0, t // 4, 9216 + (t % 4) * 55296: 9216 + (t % 4 + 1) * 55296] = cifs[m, t]
            if bits_layout == 1:
                classed = np.empty_like(hist)
                classed[:, :, idx] = hist
                hist = classed
            d_hist = torch.from_numpy(np.ascontiguousarray(hist)).cuda()
            total = sum(nb for _, nb in ok)
            d_out = torch.zeros((1, 4, total), dtype=torch.uint8, device="cuda")
            d_res = torch.zeros((4 * len(subs), 16), dtype=torch.uint8, device="cuda")
            for frame in (3, 4):
                ctx.msc_decode_frames(d_hist, 1, H * dabgpu.NB_FRAME_BITS, H, frame, subs, d_out, 4 * total, d_res, tie_rule=model, bits_layout=bits_layout)
                torch.cuda.synchronize()
                out = d_out.cpu().numpy()[0]
                off = 0
                for c, nb in ok:
                    for cc in (range(3, 4) if frame == 3 else range(4)):
                        want = fx["msc"][model][c["index"]][4 * frame + cc]
                        assert len(want) == nb and out[cc, off:off + nb].tobytes() == want, (c["name"], frame, cc)
                        n_checked += 1
                    off += nb
    finally:
        ctx.viterbi_set_mapping(0)
    assert refused == ["uep_34"] and n_checked == 5 * (len(cs) - 1)
