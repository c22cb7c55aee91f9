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
def test_mode_frames_match_oracle(ctx, oracle, mode, spb):
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
    ctx.ofdm_demod_frames_mode(mode, d_iq, n_frames, d_bits, freq_offset=d_freq, cp_corr=d_corr, fft=d_fft, symbols_per_block=spb)
    ctx.ofdm_phase_update_mode(mode, d_corr, n_frames, total_phase=d_total, fine_freq=d_fine, beta=0.9)
    torch.cuda.synchronize()
    bits, corr, fft = d_bits.cpu().numpy(), d_corr.cpu().numpy(), d_fft.cpu().numpy()
    total, fine = d_total.cpu().numpy(), d_fine.cpu().numpy()
    for k in range(n_frames):
        r = oracle.demod_frame_mode(mode, frames[k], float(freqs[k]), want_fft=True)
        assert np.array_equal(bits[k], r["bits"]), (mode, k)
        assert np.array_equal(u32(corr[k]).reshape(-1), u32(r["cp_corr"]).reshape(-1)), (mode, k)
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
