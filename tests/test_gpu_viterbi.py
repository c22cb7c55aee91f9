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


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_fic_frames_match_oracle(ctx, oracle, tie_rule):
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


def test_edge_inputs(ctx, oracle):
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


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_msc_frames_with_history_ring(ctx, oracle, tie_rule):
    """3 ensembles x mixed multiplex (EEP-A, EEP-B, 2-A special, UEP) through 7 frames (28 CIFs): the kernel's
    de-interleave-by-index over the frame-history ring must equal CIF_Deinterleaver + MSC_Decoder of the oracle"""
    import dabgpu
    import torch
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
        hist[:, slot].copy_(torch.from_numpy(frame).cuda())
        ctx.msc_decode_frames(hist, n_ens, H * oracle.NB_FRAME_BITS, H, slot, gsubs, d_out, 4 * cif_out, d_res, tie_rule=tie_rule)
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
