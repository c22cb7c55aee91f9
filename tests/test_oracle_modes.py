"""CPU tests of the mode-generic oracle functions (transmission modes II-IV): geometry and carrier mapper
against the reference objects' output (golden vectors), the generic FFT / frame demodulator against the mode I functions
(bit-identical) and against DFT known answers, the PLL with the reference's scalar tail, TX -> RX bit identity."""
import numpy as np
import pytest


def test_geometry_and_mapper_match_reference(oracle, golden):
    # golden ofdm_params_mode1 is {frame symbols, symbol period, null period, cyclic prefix, fft, carriers} from get_DAB_OFDM_params(1)
    g = oracle.geometry(1)
    assert [g.nb_frame_symbols, g.nb_symbol_period, g.nb_null_period, g.nb_cp, g.nb_fft, g.nb_carriers] == [int(v) for v in golden["ofdm_params_mode1"]]
    R = oracle.ref()
    for mode in (1, 2, 3, 4):
        g = oracle.geometry(mode)
        m = oracle.mapper_n(g.nb_fft, g.nb_carriers)
        assert sorted(m) == list(range(g.nb_carriers))
        if R is not None:                                   # the reference's own tables, compiled in place
            op = np.zeros(6, np.uint64)
            R.ref_get_ofdm_params(mode, op.ctypes.data)
            assert [g.nb_frame_symbols, g.nb_symbol_period, g.nb_null_period, g.nb_cp, g.nb_fft, g.nb_carriers] == [int(v) for v in op]
            mr = np.zeros(g.nb_carriers, np.int32)
            R.ref_get_mapper(mr.ctypes.data, g.nb_carriers, g.nb_fft)
            assert np.array_equal(m, mr)
    assert np.array_equal(oracle.mapper_n(2048, 1536), golden["mapper"])
    for mode in (2, 3, 4):                                      # the same tables as committed data (no reference tree needed)
        g = oracle.geometry(mode)
        assert [g.nb_frame_symbols, g.nb_symbol_period, g.nb_null_period, g.nb_cp, g.nb_fft, g.nb_carriers] == \
               [int(v) for v in golden[f"ofdm_params_mode{mode}"]]
        assert np.array_equal(oracle.mapper_n(g.nb_fft, g.nb_carriers), golden[f"mapper_mode{mode}"])
        assert np.array_equal(oracle.prs_fft_mode(mode).view(np.uint32), golden[f"prs_fft_mode{mode}"].view(np.uint32))
    assert np.array_equal(oracle.prs_fft_mode(1).view(np.uint32), golden["prs_fft"].view(np.uint32))


def test_product_host_tables_for_all_modes(oracle):
    import dabgpu
    for mode in (1, 2, 3, 4):
        g, p = oracle.geometry(mode), dabgpu.ofdm_params(mode)
        assert [p[k] for k in ("nb_frame_symbols", "nb_symbol_period", "nb_null_period", "nb_fft", "nb_cyclic_prefix", "nb_data_carriers",
                               "nb_frame_samples", "nb_sym_bits", "nb_frame_bits")] == \
               [g.nb_frame_symbols, g.nb_symbol_period, g.nb_null_period, g.nb_fft, g.nb_cp, g.nb_carriers, g.nb_frame_samples,
                g.nb_sym_bits, g.nb_frame_bits]
        assert np.array_equal(dabgpu.carrier_mapper(mode), oracle.mapper_n(g.nb_fft, g.nb_carriers))
        prs = np.zeros(2 * g.nb_fft, np.float32)
        dabgpu.check(dabgpu.lib().dabgpu_get_prs_fft_ref(mode, prs.ctypes.data), "get_prs_fft_ref")
        assert np.array_equal(prs.view(np.uint32), oracle.prs_fft_mode(mode).view(np.uint32))
    with pytest.raises(dabgpu.DabGpuError):
        dabgpu.ofdm_params(5)


def test_generic_functions_reproduce_mode_1(oracle):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    assert np.array_equal(oracle.fft_n(x).view(np.uint32), oracle.fft2048(x).view(np.uint32))
    assert np.array_equal(oracle.fft_n(x, True).view(np.uint32), oracle.fft2048(x, True).view(np.uint32))
    fr = (rng.standard_normal(oracle.NB_FRAME_SAMPLES) + 1j * rng.standard_normal(oracle.NB_FRAME_SAMPLES)).astype(np.complex64)
    a, b = oracle.demod_frame(fr, 1.3e-4, want_fft=True), oracle.demod_frame_mode(1, fr, 1.3e-4, want_fft=True)
    assert np.array_equal(a["bits"], b["bits"]) and np.array_equal(a["fft"].view(np.uint32), b["fft"].view(np.uint32))
    assert np.array_equal(a["cp_corr"].view(np.uint32), b["cp_corr"].view(np.uint32)) and a["total_phase"] == b["total_phase"]
    assert oracle.update_fine_freq_mode(1, 1e-5, 0.3) == oracle.update_fine_freq(1e-5, 0.3)


@pytest.mark.parametrize("n", [256, 512, 1024])
def test_generic_fft_known_answers(oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(oracle.fft_n(x) - ref).max() <= 4e-7 * np.abs(ref).max() * np.log2(n)
    assert np.abs(oracle.fft_n(x, True) - np.fft.ifft(x.astype(np.complex128)) * n).max() <= 4e-7 * np.abs(ref).max() * np.log2(n)
    e = np.zeros(n, np.complex64); e[3] = 1
    assert np.abs(oracle.fft_n(e) - np.exp(-2j * np.pi * 3 * np.arange(n) / n)).max() < 3e-7


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_tx_rx_bit_identity_and_cfo(oracle, mode):
    import modes_model as MM
    rng = np.random.default_rng(mode)
    g = oracle.geometry(mode)
    bits = rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8)
    frame = MM.make_frame(oracle, mode, bits, rng)
    r = oracle.demod_frame_mode(mode, frame, 0.0)
    assert np.array_equal((r["bits"] >= 0).astype(np.uint8), bits)
    assert np.abs(r["cp_phase"]).max() < 1e-3
    # a carrier offset undone by the PLL (incl. the scalar tail samples of every symbol) leaves the bits intact
    f = np.float32(7.0e-4)
    shifted = oracle.apply_pll(frame, -f, 0.0)
    noisy = (shifted + 0.05 * (rng.standard_normal(frame.size) + 1j * rng.standard_normal(frame.size))).astype(np.complex64)
    r2 = oracle.demod_frame_mode(mode, noisy, f)
    assert np.array_equal((r2["bits"] >= 0).astype(np.uint8), bits)
    # an uncorrected small offset shows up as the cyclic-prefix phase: 2 pi f N per symbol
    r3 = oracle.demod_frame_mode(mode, oracle.apply_pll(frame, np.float32(2e-5), 0.0), 0.0)
    assert abs(np.mean(r3["cp_phase"]) - 2 * np.pi * 2e-5 * g.nb_fft) < 2e-3
