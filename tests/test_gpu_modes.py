"""-m gpu parity tests of the size-generic OFDM demodulation kernel (transmission modes II, III, IV)
through the C ABI against the CPU oracle: soft bits byte for byte, cyclic-prefix correlations, spectra and the
fine-frequency update as float32 BIT PATTERNS; mode I through the generic kernel must equal the register-resident mode I
kernel."""
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


@pytest.mark.parametrize("mode", [2, 3, 4])
@pytest.mark.parametrize("spb", [0, 7])
@pytest.mark.parametrize("views", [True, False], ids=["with_fft_view", "bits_only"])
def test_mode_frames_match_oracle(ctx, oracle, mode, spb, views):
    """with the FFT view every mode runs the size-generic LDS kernel; without it modes II and IV run the register-resident
    wave kernel (csrc/ofdm_wave512.hip): both against the oracle, bit for bit"""
    import dabgpu
    import modes_model as MM
    import torch
    rng = np.random.default_rng(10 * mode + spb)
    g = oracle.geometry(mode)
    n_frames = 3
    frames, freqs, sent = [], [], []
    for k in range(n_frames):
        bits = rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8)
        fr = MM.make_frame(oracle, mode, bits, rng)
        f = np.float32([0.0, 6.1e-4, -2.3e-3][k])
        fr = oracle.apply_pll(fr, -f, 0.25)
        fr = (fr + 0.08 * (rng.standard_normal(fr.size) + 1j * rng.standard_normal(fr.size))).astype(np.complex64)
        frames.append(fr); freqs.append(f); sent.append(bits)
    frames = np.stack(frames)
    d_iq = torch.view_as_real(torch.from_numpy(frames).cuda())
    d_freq = torch.from_numpy(np.array(freqs, np.float32)).cuda()
    d_bits = torch.zeros((n_frames, g.nb_frame_bits), dtype=torch.int8, device="cuda")
    d_corr = torch.zeros((n_frames, g.nb_frame_symbols, 2), dtype=torch.float32, device="cuda")
    d_fft = torch.zeros((n_frames, g.nb_frame_symbols + 1, g.nb_fft, 2), dtype=torch.float32, device="cuda")
    d_total = torch.zeros(n_frames, dtype=torch.float32, device="cuda")
    d_fine = torch.from_numpy(np.array([0.0, 1e-5, -2e-5], np.float32)).cuda()
    ctx.ofdm_demod_frames_mode(mode, d_iq, n_frames, d_bits, freq_offset=d_freq, cp_corr=d_corr, fft=d_fft if views else None, symbols_per_block=spb)
    ctx.ofdm_phase_update_mode(mode, d_corr, n_frames, total_phase=d_total, fine_freq=d_fine, beta=0.9)
    torch.cuda.synchronize()
    bits, corr, fft = d_bits.cpu().numpy(), d_corr.cpu().numpy(), d_fft.cpu().numpy()
    total, fine = d_total.cpu().numpy(), d_fine.cpu().numpy()
    for k in range(n_frames):
        r = oracle.demod_frame_mode(mode, frames[k], float(freqs[k]), want_fft=True)
        assert np.array_equal(bits[k], r["bits"]), (mode, k)
        assert np.array_equal(u32(corr[k]).reshape(-1), u32(r["cp_corr"]).reshape(-1)), (mode, k)
        if views:
            assert np.array_equal(u32(fft[k]).reshape(-1), u32(r["fft"]).reshape(-1)), (mode, k)
        assert u32(total[k:k + 1])[0] == u32(np.array([r["total_phase"]], np.float32))[0]
        exp_fine = oracle.update_fine_freq_mode(mode, [0.0, 1e-5, -2e-5][k], r["total_phase"], 0.9)
        assert u32(fine[k:k + 1])[0] == u32(np.array([exp_fine], np.float32))[0]
        assert np.array_equal((bits[k] >= 0).astype(np.uint8), sent[k]), "hard bits must be the transmitted bits"


def test_mode_1_through_the_generic_kernel_equals_the_mode_1_kernel(ctx, oracle):
    import dabgpu
    import torch
    rng = np.random.default_rng(3)
    n = 2
    frames = np.stack([oracle.tx_to_frame_buffer(oracle.apply_pll(np.concatenate([oracle.modulate_frame(
        rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8))] * 2), 3e-4 * (k + 1), 0.0)) for k in range(n)])
    d_iq = torch.view_as_real(torch.from_numpy(frames).cuda())
    d_freq = torch.tensor([-3e-4, -6e-4], dtype=torch.float32, device="cuda")
    outs = []
    for generic in (False, True):
        d_bits = torch.zeros((n, oracle.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
        d_corr = torch.zeros((n, 76, 2), dtype=torch.float32, device="cuda")
        d_fft = torch.zeros((n, 77, 2048, 2), dtype=torch.float32, device="cuda")
        if generic:
            ctx.ofdm_demod_frames_mode(1, d_iq, n, d_bits, freq_offset=d_freq, cp_corr=d_corr, fft=d_fft)
        else:
            ctx.ofdm_demod_frames(d_iq, d_bits, freq_offset=d_freq, cp_corr=d_corr, fft=d_fft, n_frames=n)
        torch.cuda.synchronize()
        outs.append((d_bits.cpu().numpy(), u32(d_corr.cpu().numpy()), u32(d_fft.cpu().numpy())))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))


def test_invalid_mode_is_rejected(ctx):
    import dabgpu
    import torch
    x = torch.zeros(16, dtype=torch.float32, device="cuda")
    b = torch.zeros(16, dtype=torch.int8, device="cuda")
    with pytest.raises(dabgpu.DabGpuError):
        ctx.ofdm_demod_frames_mode(5, x, 1, b)


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_prs_sync_of_other_modes_matches_oracle(ctx, oracle, mode):
    """coarse frequency + fine time synchronisation on the mode's own phase reference symbol: state and both dB responses"""
    import dabgpu
    import modes_model as MM
    rng = np.random.default_rng(50 + mode)
    g = oracle.geometry(mode)
    prs_o = oracle.prs_fft_mode(mode)
    assert np.array_equal(u32(prs_o), u32(np.frombuffer(_prs_bytes(dabgpu, mode, g.nb_fft), np.complex64)))
    tx = np.concatenate([MM.make_tx_frame(oracle, mode, rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8), rng) for _ in range(2)])
    for cfo_bins, toff, noise in [(0.0, 0, 0.0), (2.4, 17, 0.05), (-5.25, -30, 0.1)]:
        x = oracle.apply_pll(tx, cfo_bins / g.nb_fft, 0.1)
        x = (x + noise * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64) / np.float32(39.2)
        start = g.nb_null_period - toff
        sym = x[start:start + g.nb_fft].copy()
        ost = oracle.SyncState(0.0, 0.0, 0, 0, 0, 0)
        gst = dabgpu.SyncState(0.0, 0.0, 0, 0, 0, 0)
        cfg_o = oracle.sync_cfg_default(); cfg_o.impulse_peak_threshold_db = 8.0
        cfg_g = dabgpu.sync_cfg_default(); cfg_g.impulse_peak_threshold_db = 8.0
        for it in range(2):                                  # second pass exercises the slow-beta branch of the coarse loop
            resp = oracle.coarse_freq_sync_mode(mode, sym, ost, cfg_o)
            f = np.float32(np.float32(ost.freq_coarse) + np.float32(ost.freq_fine))
            ok, off, ir = oracle.fine_time_sync_mode(mode, sym, f, cfg_o)
            imp, frq = ctx.ofdm_sync_host_mode(mode, sym, gst, cfg_g)
            assert np.array_equal(u32(frq), u32(resp)) and np.array_equal(u32(imp), u32(ir)), (mode, cfo_bins, it)
            assert u32(np.float32(gst.freq_coarse))[()] == u32(np.float32(ost.freq_coarse))[()]
            assert u32(np.float32(gst.freq_fine))[()] == u32(np.float32(ost.freq_fine))[()]
            assert bool(gst.sync_valid) == ok and (not ok or gst.fine_time_offset == off), (mode, cfo_bins, it)
        assert abs(ost.freq_coarse * g.nb_fft + cfo_bins) < 1.0


def _prs_bytes(dabgpu, mode, n):
    buf = np.zeros(2 * n, np.float32)
    dabgpu.check(dabgpu.lib().dabgpu_get_prs_fft_ref(mode, buf.ctypes.data), "get_prs_fft_ref")
    return buf.tobytes()


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_cpp_mirror_stream_in_other_modes(oracle, tmp_path, mode):
    """OFDM_Demod mirror class constructed for mode II / III / IV on an unsynchronised stream vs the oracle state machine"""
    import os
    import subprocess
    import modes_model as MM
    import stream_model as SM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    harness = os.path.join(root, "tests", "cpp", "mirror_harness")
    if not os.path.exists(harness):
        import __graft_entry__ as gr
        gr.build()
    rng = np.random.default_rng(mode)
    g = oracle.geometry(mode)
    sent = [rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8) for _ in range(8)]
    tx = oracle.apply_pll(np.concatenate([MM.make_tx_frame(oracle, mode, b, rng) for b in sent]), 2.1 / g.nb_fft, 0.2)
    stream = np.concatenate([tx[g.nb_null_period:g.nb_null_period + 6000 + 77], tx])
    stream = ((stream + 0.05 * (rng.standard_normal(stream.size) + 1j * rng.standard_normal(stream.size))) / 39.2).astype(np.complex64)
    (tmp_path / "iq.c32").write_bytes(stream.tobytes())
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(root, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env["DABGPU_HARNESS_MODE"] = str(mode)
    env["DABGPU_HARNESS_PEAK_DB"] = "8"          # the reference's default 20 dB rarely passes with the short symbols of modes II / III
    res = subprocess.run([harness, str(tmp_path / "iq.c32"), str(out), "16384"], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    model = SM.StreamModel(oracle, mode)
    model.cfg.impulse_peak_threshold_db = 8.0
    for k in range(0, stream.size, 16384):
        model.process(stream[k:k + 16384])
    nf = len(model.out_frames)
    assert nf >= 4
    assert f"frames={nf} read={nf} desync={model.frames_desync} state={model.state}" in res.stdout, res.stdout
    bits = np.fromfile(out / "frame_bits.bin", dtype=np.int8).reshape(nf, g.nb_frame_bits)
    states = np.fromfile(out / "states.bin", dtype=np.float32).reshape(nf, 4)
    fft1 = np.fromfile(out / "fft_sym1.bin", dtype=np.complex64).reshape(nf, g.nb_fft)
    good = 0
    for k, fr in enumerate(model.out_frames):
        assert np.array_equal(bits[k], fr["bits"]), (mode, k)
        assert u32(states[k, 0:1])[0] == u32(np.array([fr["coarse"]], np.float32))[0]
        assert u32(states[k, 1:2])[0] == u32(np.array([fr["fine"]], np.float32))[0]
        assert int(states[k, 2]) == fr["offset"] and int(states[k, 3]) == fr["desync"]
        assert np.array_equal(u32(fft1[k]), u32(fr["fft"][g.nb_fft:2 * g.nb_fft]))
        good += any(np.array_equal((bits[k] >= 0).astype(np.uint8), b) for b in sent)
    # the reference's loop settles within a few frames in modes III and IV; with mode II's 512-point PRS its coarse / fine
    # estimates keep jittering on this noisy stream (same behaviour in the oracle: parity is the criterion there)
    # (a statement about this seed's noise, not about parity: under DAB_FUZZ_OFFSET other realisations of mode III need more than the 8 frames)
    if mode != 2 and not os.environ.get("DAB_FUZZ_OFFSET"):
        assert good >= 1, "once locked, the hard bits are the transmitted bits"


@pytest.mark.parametrize("spb", [0, 1, 2, 6, 7, 152])
def test_mode_3_pairs_of_symbols_equal_one_symbol_per_wavefront(ctx, spb):
    """mode III demodulates two symbols per wavefront (csrc/ofdm_wave512.hip ofdm_demod_wave3_kernel); DABGPU_MODE3_SINGLE=1 selects
    the one-symbol-per-wavefront path it replaced: soft bits and correlations must be identical for even and odd runs of symbols
    (runs of 1, 2, 6, 7, 19 and the whole frame of 152 data symbols)"""
    import os
    import torch
    rng = np.random.default_rng(33 + spb)
    n, fs = 5, 153 * 319 + 345
    iq = torch.from_numpy(rng.standard_normal((n, fs, 2)).astype(np.float32)).cuda()
    freq = torch.from_numpy(((rng.random(n) * 2 - 1) * 3.0e-3).astype(np.float32)).cuda()
    out = {}
    for single in (False, True):
        if single:
            os.environ["DABGPU_MODE3_SINGLE"] = "1"
        try:
            bits = torch.zeros((n, 152 * 384), dtype=torch.int8, device="cuda")
            corr = torch.zeros((n, 153, 2), dtype=torch.float32, device="cuda")
            ctx.ofdm_demod_frames_mode(3, iq, n, bits, freq_offset=freq, cp_corr=corr, symbols_per_block=spb)
            torch.cuda.synchronize()
            out[single] = (bits.cpu().numpy(), corr.cpu().numpy().view(np.uint32))
        finally:
            os.environ.pop("DABGPU_MODE3_SINGLE", None)
    assert len(np.unique(out[False][0])) > 100
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
