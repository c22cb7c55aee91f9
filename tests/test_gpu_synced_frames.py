"""GPU parity test of dabgpu_ofdm_sync_demod_frames -- PRS synchronisation, the demodulation it positions and corrects, and the
fine-frequency update as ONE call with the sync records staying on the device -- against the CPU oracle composed the way
OFDM_Demod runs a frame (ofdm_demodulator.cpp:360-548 -> :650-766 -> :606-618): soft bits byte for byte,
frequency offsets and the total phase as float32 BIT PATTERNS, time offsets and flags as integers, over several consecutive frames
(the tracked state feeds the next frame's synchroniser: fast and slow coarse updates, the fine loop's wrap)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 700                               # expected PRS position inside a receiver's slice (>= 504)
STRIDE = P + 1544 + 196608            # the oracle's demod_frame wants the whole frame buffer incl. the trailing NULL


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


def make_slices(oracle, n_frames, cases, seed):
    """[n_frames][n][STRIDE] complex64: receiver k's frame j starts at sample P + toff_k of its slice; carrier offset cfo_k (bins), noise;
    a case with cfo None is noise only (its impulse-peak test fails)"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_frames, len(cases), STRIDE), np.complex64)
    for k, (cfo, toff, noise) in enumerate(cases):
        for j in range(n_frames):
            if cfo is None:
                out[j, k] = (rng.standard_normal(STRIDE) + 1j * rng.standard_normal(STRIDE)).astype(np.complex64)
                continue
            bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
            tx = np.concatenate([oracle.modulate_frame(bits)] * 2)                    # NULL, frame, NULL, frame
            tx = oracle.apply_pll(tx, cfo / 2048.0, 0.05 * j)
            start = oracle.NB_NULL_PERIOD - (P + toff)                                # slice sample P + toff = first PRS sample
            assert start >= 0
            seg = tx[start:start + STRIDE]
            out[j, k, :seg.size] = seg
            if noise > 0:
                out[j, k] += (noise * (rng.standard_normal(STRIDE) + 1j * rng.standard_normal(STRIDE))).astype(np.complex64)
    return out


def oracle_frames(oracle, slices, cfg):
    """the oracle composed like the reference: per receiver a persistent (coarse, fine, found) record"""
    n_frames, n, _ = slices.shape
    conj_ref, time_ref = oracle.sync_refs()
    states = [oracle.SyncState(0.0, 0.0, 0, 0, 0, 0) for _ in range(n)]
    exp = []
    for j in range(n_frames):
        row = []
        for k in range(n):
            st = states[k]
            prs_sym = slices[j, k, P:P + 2048]
            oracle.coarse_freq_sync(prs_sym, st, cfg, time_ref)
            f = np.float32(np.float32(st.freq_coarse) + np.float32(st.freq_fine))
            ok, off, _ = oracle.fine_time_sync(prs_sym, f, cfg, conj_ref)
            rec = dict(ok=ok, off=off, coarse=np.float32(st.freq_coarse))
            if ok:
                r = oracle.demod_frame(slices[j, k, P + off:P + off + oracle.NB_FRAME_SAMPLES], f)
                st.freq_fine = float(oracle.update_fine_freq(st.freq_fine, r["total_phase"]))
                rec.update(bits=r["bits"], total=r["total_phase"], corr=r["cp_corr"])
            rec["fine"] = np.float32(st.freq_fine)
            row.append(rec)
        exp.append(row)
    return exp


CASES = [(0.0, 0, 0.0), (3.7, 37, 4.0), (-11.25, -81, 8.0), (0.45, 201, 2.0), (None, 0, 0.0), (7.5, -100, 1.0), (-0.3, 99, 6.0)]


@pytest.mark.parametrize("layout,spb", [(0, 25), (0, 75), (1, 0), (1, 38)])
def test_synced_frames_match_the_oracle_composition(ctx, oracle, layout, spb):
    import dabgpu
    import torch
    n_frames, n = 3, len(CASES)
    slices = make_slices(oracle, n_frames, CASES, seed=77)
    cfg_o = oracle.sync_cfg_default()
    exp = oracle_frames(oracle, slices, cfg_o)
    sdt = np.dtype(dabgpu.SYNC_STATE_DTYPE)
    d_st = torch.from_numpy(np.zeros(n, sdt).view(np.uint8)).cuda()
    nat = dabgpu.classed_to_natural_index() if layout else None
    for j in range(n_frames):
        d_iq = torch.from_numpy(slices[j].view(np.float32)).cuda()
        bits = torch.full((n, 230400), 55, dtype=torch.int8, device="cuda")
        corr = torch.full((n, 76, 2), 7.0, dtype=torch.float32, device="cuda")
        total = torch.full((n,), -3.0, dtype=torch.float32, device="cuda")
        ctx.ofdm_sync_demod_frames(d_iq, n, STRIDE, P, d_st, bits, cp_corr=corr, symbols_per_block=spb, bits_layout=layout, total_phase=total)
        torch.cuda.synchronize()
        got = d_st.cpu().numpy().view(sdt)
        gb, gc, gt = bits.cpu().numpy(), corr.cpu().numpy(), total.cpu().numpy()
        for k in range(n):
            e = exp[j][k]
            what = f"frame {j}, receiver {k} {CASES[k]}"
            assert bool(got[k]["sync_valid"]) == e["ok"], what
            assert u32(got[k]["freq_coarse"]) == u32(e["coarse"]), what
            assert u32(got[k]["freq_fine"]) == u32(e["fine"]), what
            assert got[k]["is_found_coarse"] == 1
            if not e["ok"]:
                assert (gb[k] == 55).all() and (gc[k] == 7.0).all() and gt[k] == -3.0, what + ": a failed impulse-peak test must leave the rows untouched"
                continue
            assert got[k]["fine_time_offset"] == e["off"], what
            b = gb[k][nat] if layout else gb[k]
            assert np.array_equal(b, e["bits"]), what
            assert np.array_equal(u32(gc[k].reshape(-1)), u32(e["corr"].view(np.float32))), what
            assert u32(gt[k]) == u32(e["total"]), what
    # the receivers with a signal found their frame where it was put, and track the carrier to within the fine loop's reach
    for k, (cfo, toff, _) in enumerate(CASES):
        if cfo is None:
            assert not exp[-1][k]["ok"]
            continue
        assert exp[-1][k]["ok"] and exp[-1][k]["off"] == toff
        assert abs(float(exp[-1][k]["coarse"]) + float(exp[-1][k]["fine"]) + cfo / 2048.0) < 0.6 / 2048.0


def test_synced_frames_equal_the_three_separate_calls(ctx, oracle):
    """dabgpu_ofdm_sync + dabgpu_ofdm_demod_frames_history on the shifted frames + dabgpu_ofdm_phase_update, composed on the host,
    give the same bits / states (the one-call form only keeps the records on the device); ring stride and a NULL correlation buffer"""
    import dabgpu
    import torch
    cases = [(2.2, 12, 3.0), (-5.4, -33, 5.0), (0.0, 1, 0.5)]
    n = len(cases)
    slices = make_slices(oracle, 1, cases, seed=5)[0]
    sdt = np.dtype(dabgpu.SYNC_STATE_DTYPE)
    d_iq = torch.from_numpy(slices.view(np.float32)).cuda()
    fmt = dabgpu.IQ_FORMATS.index("raw_f32l")
    # separate calls
    d_st = torch.from_numpy(np.zeros(n, sdt).view(np.uint8)).cuda()
    ctx.ofdm_sync(d_iq.view(-1)[2 * P:], n, STRIDE, d_st)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy().view(sdt).copy()
    assert st["sync_valid"].all()
    frames = np.stack([slices[k, P + st[k]["fine_time_offset"]:P + st[k]["fine_time_offset"] + 196608] for k in range(n)])
    d_frames = torch.from_numpy(frames.view(np.float32)).cuda()
    f = torch.from_numpy((st["freq_coarse"] + st["freq_fine"]).astype(np.float32)).cuda()
    stride = 2 * 230400
    ring_a = torch.zeros((n, 2, 230400), dtype=torch.int8, device="cuda")
    corr = torch.zeros((n, 76, 2), dtype=torch.float32, device="cuda")
    ctx.ofdm_demod_frames_history(d_frames, fmt, n, ring_a[:, 1], freq_offset=f, cp_corr=corr, bits_frame_stride=stride)
    fine = torch.from_numpy(st["freq_fine"].copy()).cuda()
    total_a = torch.zeros(n, dtype=torch.float32, device="cuda")
    ctx.ofdm_phase_update(corr, n, total_phase=total_a, fine_freq=fine, beta=0.9)
    # one call
    d_st2 = torch.from_numpy(np.zeros(n, sdt).view(np.uint8)).cuda()
    ring_b = torch.zeros((n, 2, 230400), dtype=torch.int8, device="cuda")
    total_b = torch.zeros(n, dtype=torch.float32, device="cuda")
    ctx.ofdm_sync_demod_frames(d_iq, n, STRIDE, P, d_st2, ring_b[:, 1], bits_frame_stride=stride, total_phase=total_b)
    torch.cuda.synchronize()
    st2 = d_st2.cpu().numpy().view(sdt)
    assert torch.equal(ring_a, ring_b) and ring_b[:, 0].abs().sum().item() == 0 and ring_b[:, 1].abs().sum().item() > 0
    assert torch.equal(total_a.view(torch.int32), total_b.view(torch.int32))
    assert np.array_equal(u32(st2["freq_fine"]), u32(fine.cpu().numpy())) and np.array_equal(u32(st2["freq_coarse"]), u32(st["freq_coarse"]))
    assert np.array_equal(st2["fine_time_offset"], st["fine_time_offset"])


def test_synced_frames_argument_checks(ctx):
    import dabgpu
    import torch
    n = 2
    d_iq = torch.zeros((n, STRIDE, 2), dtype=torch.float32, device="cuda")
    d_st = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    bits = torch.zeros((n, 230400), dtype=torch.int8, device="cuda")
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_sync_demod_frames(d_iq, n, STRIDE, 100, d_st, bits)                    # frame could start before the slice
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_sync_demod_frames(d_iq, n, P + 76 * 2552, P, d_st, bits)              # slice shorter than the latest possible frame
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_sync_demod_frames(d_iq, n, STRIDE, P, d_st, bits, bits_layout=7)
    ctx.ofdm_sync_demod_frames(d_iq, 0, STRIDE, P, d_st, bits)                         # empty batch: nothing to do
    # all-zero input: every dB value is -inf, peak - mean is NaN, and the reference's test `peak - mean < threshold` is false for NaN: the
    # frame counts as found at the first sample (:503, :529) -- and demodulates to soft bits 0
    ctx.ofdm_sync_demod_frames(d_iq, n, STRIDE, P, d_st, bits)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy().view(np.dtype(dabgpu.SYNC_STATE_DTYPE))
    assert st["sync_valid"].all() and (st["fine_time_offset"] == -504).all() and bits.abs().sum().item() == 0
