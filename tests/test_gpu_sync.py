"""GPU parity tests of the PRS synchronisation kernel (coarse frequency + fine time) through the C ABI against the
CPU oracle: coarse/fine offsets and both dB responses compared as float32 BIT PATTERNS, offsets and flags as integers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def make_stream(oracle, seed, cfo_bins, toff, noise):
    """two back-to-back frames with CFO, returns the 2048 samples at the expected PRS position shifted by toff"""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
    tx = np.concatenate([oracle.modulate_frame(bits)] * 2)
    tx = oracle.apply_pll(tx, cfo_bins / 2048.0, 0.1)
    if noise > 0:
        tx = (tx + noise * (rng.standard_normal(tx.size) + 1j * rng.standard_normal(tx.size))).astype(np.complex64)
    start = oracle.NB_NULL_PERIOD - toff          # corr buffer position nb_null_period when the NULL was found toff late
    return tx[start:start + 2048].copy(), tx


@pytest.mark.parametrize("cfo_bins,toff,noise", [(0.0, 0, 0.0), (3.7, 37, 4.0), (-11.25, -80, 8.0), (0.45, 200, 2.0)])
def test_sync_matches_oracle_over_frames(ctx, oracle, cfo_bins, toff, noise):
    import dabgpu
    prs_sym, _ = make_stream(oracle, 11, cfo_bins, toff, noise)
    conj_ref, time_ref = oracle.sync_refs()
    ost = oracle.SyncState(0.0, 0.0, 0, 0, 0, 0)
    gst = dabgpu.SyncState(0.0, 0.0, 0, 0, 0, 0)
    cfg_o = oracle.sync_cfg_default()
    for it in range(3):                       # first call = fast update, later calls = slow (beta 0.1) path
        fr_o = oracle.coarse_freq_sync(prs_sym, ost, cfg_o, time_ref)
        f = np.float32(np.float32(ost.freq_coarse) + np.float32(ost.freq_fine))
        ok_o, off_o, ir_o = oracle.fine_time_sync(prs_sym, f, cfg_o, conj_ref)
        gst, ir_g, fr_g = ctx.ofdm_sync_host(prs_sym, gst)
        assert np.array_equal(u32(fr_g), u32(fr_o)), f"coarse response, iteration {it}"
        assert np.array_equal(u32(ir_g), u32(ir_o)), f"impulse response, iteration {it}"
        assert u32(np.float32(gst.freq_coarse)) == u32(np.float32(ost.freq_coarse))
        assert u32(np.float32(gst.freq_fine)) == u32(np.float32(ost.freq_fine))
        assert gst.is_found_coarse == 1 and bool(gst.sync_valid) == ok_o
        if ok_o:
            assert gst.fine_time_offset == off_o
    assert ok_o and gst.fine_time_offset == toff
    # the coarse estimator is an integer-bin search with a 3-point interpolation: within one bin of the true offset
    assert abs(float(gst.freq_coarse) + float(gst.freq_fine) + cfo_bins / 2048.0) < 1.0 / 2048.0


def test_sync_rejects_noise_and_handles_zero_input(ctx, oracle):
    import dabgpu
    rng = np.random.default_rng(5)
    conj_ref, time_ref = oracle.sync_refs()
    for x in ((rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64),
              np.zeros(2048, np.complex64)):
        ost = oracle.SyncState(0.0, 0.0, 0, 0, 0, 0)
        oracle.coarse_freq_sync(x, ost, None, time_ref)
        f = np.float32(np.float32(ost.freq_coarse) + np.float32(ost.freq_fine))
        ok_o, off_o, ir_o = oracle.fine_time_sync(x, f, None, conj_ref)
        gst, ir_g, _ = ctx.ofdm_sync_host(x, dabgpu.SyncState(0.0, 0.0, 0, 0, 0, 0))
        assert bool(gst.sync_valid) == ok_o
        assert np.array_equal(u32(ir_g), u32(ir_o))
        assert u32(np.float32(gst.freq_coarse)) == u32(np.float32(ost.freq_coarse)) or (
            np.isnan(gst.freq_coarse) and np.isnan(ost.freq_coarse))


def test_sync_batch_and_coarse_disabled(ctx, oracle):
    import dabgpu
    import torch
    n = 5
    syms = np.stack([make_stream(oracle, 20 + k, 1.3 * k - 2.0, 10 * k, 3.0)[0] for k in range(n)])
    d_syms = torch.from_numpy(syms.view(np.float32)).cuda()
    st0 = np.zeros(n, dtype=np.dtype(dabgpu.SYNC_STATE_DTYPE))
    st0["freq_fine"] = np.linspace(-1e-4, 1e-4, n)
    d_st = torch.from_numpy(st0.view(np.uint8)).cuda()
    cfg = dabgpu.sync_cfg_default()
    cfg.is_coarse_freq_correction = 0
    ctx.ofdm_sync(d_syms, n, 2048, d_st, cfg=cfg)
    torch.cuda.synchronize()
    got = d_st.cpu().numpy().view(np.dtype(dabgpu.SYNC_STATE_DTYPE))
    conj_ref, _ = oracle.sync_refs()
    cfg_o = oracle.sync_cfg_default()
    cfg_o.is_coarse_freq_correction = 0
    for k in range(n):
        ok_o, off_o, _ = oracle.fine_time_sync(syms[k], st0["freq_fine"][k], cfg_o, conj_ref)      # coarse forced to 0 (:363-367)
        assert got[k]["freq_coarse"] == 0.0 and bool(got[k]["sync_valid"]) == ok_o
        if ok_o:
            assert got[k]["fine_time_offset"] == off_o
