"""CPU tests (no GPU): the oracle's restatement of the reference's file-format helpers against
(a) golden vectors produced by the reference's own headers (tests/golden/make_golden_io.py) and (b) those headers
themselves when oracle/_ref is present; plus the product's host-only wav header walk and mode-string table, which
need no device.  Floats are compared as uint32 BIT PATTERNS."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gio():
    return np.load(os.path.join(ROOT, "tests", "golden", "io_format_vectors.npz"))


def test_oracle_raw_modes_match_reference_vectors(oracle, gio):
    for fmt in range(14):
        mode = oracle.IQ_MODES[fmt]
        y = oracle.iq_convert(gio[f"{mode}_in"], fmt)
        assert np.array_equal(y.view(np.uint32), gio[f"{mode}_out_u32"]), mode


def test_oracle_wav_modes_match_reference_vectors(oracle, gio):
    for fmt in range(14, 22):
        name = oracle.IQ_MODES[fmt]
        img = gio[f"{name}_image"]
        h = oracle.wav_parse_header(img)
        ref_h = gio[f"{name}_header"]
        assert h is not None and h["format"] == fmt
        assert [h["code"], h["channels"], h["rate"], h["bits"], h["data_size"], h["data_offset"]] == [int(v) for v in ref_h[1:]]
        payload = img[h["data_offset"]:h["data_offset"] + h["data_size"]]
        y = oracle.iq_convert(payload, fmt)
        assert np.array_equal(y.view(np.uint32), gio[f"{name}_out_u32"]), name


def test_wav_header_accept_reject_table(oracle, gio):
    import dabgpu
    for name, accept in zip(gio["hdr_names"], gio["hdr_accept"]):
        img = gio[f"hdr_{name}_image"]
        h = oracle.wav_parse_header(img)
        assert (h is not None) == bool(accept), name
        # the product's host-side parser (no device needed) agrees with the reference on every image
        if accept:
            ph = dabgpu.wav_parse_header(img)
            f = gio[f"hdr_{name}_fields"]
            assert [ph.audio_format, ph.total_channels, ph.samples_per_second, ph.bits_per_sample, ph.data_chunk_size,
                    ph.data_chunk_offset] == [int(v) for v in f[1:]], name
            assert ph.iq_format == h["format"]
        else:
            with pytest.raises(dabgpu.DabGpuError):
                dabgpu.wav_parse_header(img)


def test_product_mode_table_matches_reference_mode_list():
    import dabgpu
    # iq_read_modes (examples/app_helpers/app_iq_readers.h:107-113) minus "wav"
    modes = ["raw_u8", "raw_s8", "raw_s16l", "raw_s16b", "raw_u16l", "raw_u16b", "raw_s32l", "raw_s32b", "raw_u32l", "raw_u32b",
             "raw_f32l", "raw_f32b", "raw_f64l", "raw_f64b"]
    sizes = [2, 2, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 16, 16]
    for i, (m, s) in enumerate(zip(modes, sizes)):
        assert dabgpu.iq_format_from_mode(m) == i
        assert dabgpu.iq_format_sample_bytes(i) == s
    assert dabgpu.iq_format_from_mode("wav") == -1 and dabgpu.iq_format_from_mode("raw_s24l") == -1
    assert dabgpu.iq_format_sample_bytes(99) == 0 and dabgpu.iq_format_sample_bytes(dabgpu.IQ_FORMATS.index("wav_pcm24")) == 6


def test_oracle_bit_converters_match_reference_vectors(oracle, gio):
    assert np.array_equal(oracle.hard_bytes_to_soft_bits(gio["hard_in"]), gio["hard_to_soft"])
    assert np.array_equal(oracle.soft_bits_to_hard_bytes(gio["soft_in"]), gio["soft_to_hard"])
    # pack(unpack(x)) == x for every byte
    b = np.arange(256, dtype=np.uint8)
    assert np.array_equal(oracle.soft_bits_to_hard_bytes(oracle.hard_bytes_to_soft_bits(b)), b)


def test_oracle_against_reference_headers_in_place(oracle, tmp_path):
    """wider sweep against the reference's own reader chain when oracle/_ref was built here"""
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_iq_read_file"):
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rng = np.random.default_rng(5)
    for fmt in range(14):
        mode = oracle.IQ_MODES[fmt]
        size = oracle.iq_component_bytes(fmt)
        n_samples = 4099
        raw = rng.integers(0, 256, 2 * n_samples * size, dtype=np.uint8)
        if "f64" in mode or "f32" in mode:          # keep clear of signalling NaNs: x86 quiets them in the f64->f32 cast
            v = rng.standard_normal(2 * n_samples) * 10.0 ** rng.integers(-30, 30, 2 * n_samples)
            a = v.astype("<f8" if "f64" in mode else "<f4")
            raw = np.frombuffer((a.byteswap() if mode.endswith("b") else a).tobytes(), np.uint8)
        p = tmp_path / mode
        p.write_bytes(raw.tobytes())
        out = np.zeros(2 * n_samples, np.float32)
        got = R.ref_iq_read_file(str(p).encode(), mode.encode(), out.ctypes.data, n_samples, 1000)
        assert got == n_samples
        assert np.array_equal(oracle.iq_convert(raw, fmt).view(np.uint32), out.view(np.uint32)), mode
