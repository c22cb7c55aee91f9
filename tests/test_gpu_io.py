"""GPU parity tests of the data-format kernels through the C ABI: IQ sample formats -> complex float
and soft <-> hard bit packing, against the reference-generated golden vectors and the CPU oracle.  Floats are compared
as uint32 BIT PATTERNS; everything else byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import dabgpu
    c = dabgpu.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gio():
    return np.load(os.path.join(ROOT, "tests", "golden", "io_format_vectors.npz"))


def test_raw_modes_match_reference_vectors(ctx, gio):
    import dabgpu
    for fmt in range(14):
        mode = dabgpu.IQ_FORMATS[fmt]
        assert dabgpu.iq_format_from_mode(mode) == fmt
        y = ctx.iq_convert_host(gio[f"{mode}_in"], fmt)
        assert np.array_equal(y.view(np.uint32), gio[f"{mode}_out_u32"]), mode


def test_wav_files_match_reference_vectors(ctx, gio):
    import dabgpu
    for fmt in range(14, 22):
        name = dabgpu.IQ_FORMATS[fmt]
        img = gio[f"{name}_image"]
        h = dabgpu.wav_parse_header(img)
        assert h.iq_format == fmt and h.total_channels == 2
        payload = img[h.data_chunk_offset:h.data_chunk_offset + h.data_chunk_size]
        y = ctx.iq_convert_host(payload, fmt)
        assert np.array_equal(y.view(np.uint32), gio[f"{name}_out_u32"]), name


@pytest.mark.parametrize("n_samples", [1, 2, 3, 255, 1 << 20, (1 << 20) + 1])
def test_all_formats_match_oracle_ragged_sizes(ctx, oracle, n_samples):
    import dabgpu
    rng = np.random.default_rng(n_samples)
    for fmt in range(len(dabgpu.IQ_FORMATS)):
        name = dabgpu.IQ_FORMATS[fmt]
        size = oracle.iq_component_bytes(fmt)
        raw = rng.integers(0, 256, 2 * n_samples * size, dtype=np.uint8)
        if "f64" in name:                      # random bytes would hold signalling NaNs (quieted differently per ISA)
            v = rng.standard_normal(2 * n_samples) * 10.0 ** rng.integers(-50, 50, 2 * n_samples)
            raw = np.frombuffer((v.astype(">f8") if name.endswith("b") else v.astype("<f8")).tobytes(), np.uint8)
        y = ctx.iq_convert_host(raw, fmt)
        assert np.array_equal(y.view(np.uint32), oracle.iq_convert(raw, fmt).view(np.uint32)), name


def test_every_16_bit_value(ctx, oracle):
    import dabgpu
    raw = np.arange(65536, dtype="<u2")
    for name in ("raw_s16l", "raw_u16l", "raw_s16b", "raw_u16b", "wav_pcm16"):
        fmt = dabgpu.IQ_FORMATS.index(name)
        b = raw.view(np.uint8)
        assert np.array_equal(ctx.iq_convert_host(b, fmt).view(np.uint32), oracle.iq_convert(b, fmt).view(np.uint32)), name


def test_device_buffers_on_torch_stream(ctx, oracle):
    """the asynchronous entry point on torch-owned device memory: rtl_sdr style u8 capture of two whole frames"""
    import dabgpu
    import torch
    rng = np.random.default_rng(3)
    n = 2 * dabgpu.NB_FRAME_SAMPLES
    raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
    d_raw = torch.from_numpy(raw).cuda()
    d_iq = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    ctx.iq_convert(d_raw, 0, n, d_iq)
    torch.cuda.synchronize()
    assert np.array_equal(d_iq.cpu().numpy().view(np.uint32), oracle.iq_convert(raw, 0).view(np.uint32))


def test_bit_converters_match_reference_vectors(ctx, gio):
    assert np.array_equal(ctx.hard_bytes_to_soft_bits_host(gio["hard_in"]), gio["hard_to_soft"])
    assert np.array_equal(ctx.soft_bits_to_hard_bytes_host(gio["soft_in"]), gio["soft_to_hard"])


@pytest.mark.parametrize("n_bytes", [1, 2, 3, 4, 5, 7, 28800, 28801, 1 << 20])
def test_bit_converters_match_oracle_and_round_trip(ctx, oracle, n_bytes):
    rng = np.random.default_rng(n_bytes)
    soft = rng.integers(-128, 128, 8 * n_bytes, dtype=np.int8)
    hard = ctx.soft_bits_to_hard_bytes_host(soft)
    assert np.array_equal(hard, oracle.soft_bits_to_hard_bytes(soft))
    back = ctx.hard_bytes_to_soft_bits_host(hard)
    assert np.array_equal(back, oracle.hard_bytes_to_soft_bits(hard))
    assert np.array_equal(ctx.soft_bits_to_hard_bytes_host(back), hard)          # idempotent after one pass
    assert np.array_equal(back >= 0, soft >= 0)


def test_invalid_arguments_are_rejected(ctx):
    import dabgpu
    with pytest.raises(dabgpu.DabGpuError):
        ctx.iq_convert_host(np.zeros(16, np.uint8), 99)
    import torch
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(64, dtype=torch.float32, device="cuda")
    with pytest.raises(dabgpu.DabGpuError):
        ctx.iq_convert(d[1:], 0, 4, o)            # misaligned raw pointer


def quantise(frames, name):
    """complex64 frames -> capture bytes of format `name` the way an SDR front end would deliver them"""
    x = np.stack([frames.real, frames.imag], axis=-1).reshape(-1)
    x = x / np.abs(x).max()
    if name == "raw_u8":
        return np.clip(np.rint(x * 127.0 + 127.5), 0, 255).astype(np.uint8)
    if name == "raw_s8":
        return np.clip(np.rint(x * 127.0), -128, 127).astype(np.int8).view(np.uint8)
    if name in ("raw_s16l", "raw_s16b", "raw_u16l"):
        v = np.clip(np.rint(x * 30000.0), -32768, 32767).astype(np.int16)
        if name == "raw_u16l":
            return (v.astype(np.int32) + 32768).astype("<u2").view(np.uint8)
        return v.astype("<i2" if name.endswith("l") else ">i2").view(np.uint8)
    if name == "raw_f32l":
        return x.astype("<f4").view(np.uint8)
    if name == "raw_f64b":
        return x.astype(">f8").view(np.uint8)
    raise ValueError(name)


@pytest.mark.parametrize("name", ["raw_u8", "raw_s8", "raw_s16l", "raw_f32l", "raw_s16b", "raw_u16l", "raw_f64b"])
def test_demod_from_capture_format_matches_oracle(ctx, oracle, name):
    """fused raw-input loader (u8/s8/s16l/f32l) and convert-then-demod (the rest): soft bits, CP correlation and FFT
    bit-identical to oracle.iq_convert -> oracle.demod_frame, and to the product's own two-step path"""
    import dabgpu
    import torch
    rng = np.random.default_rng(17)
    n_frames = 3
    frames = []
    for k in range(n_frames):
        bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
        tx = oracle.tx_to_frame_buffer(np.concatenate([oracle.modulate_frame(bits)] * 2))
        tx = oracle.apply_pll(tx, 2.1e-4 * (k + 1), 0.2)
        frames.append((tx + 2.0 * (rng.standard_normal(tx.size) + 1j * rng.standard_normal(tx.size))).astype(np.complex64))
    frames = np.stack(frames)
    fmt = dabgpu.IQ_FORMATS.index(name)
    raw = quantise(frames, name)
    freq = np.array([-2.1e-4 * (k + 1) for k in range(n_frames)], dtype=np.float32)
    d_raw = torch.from_numpy(raw.copy()).cuda()
    d_freq = torch.from_numpy(freq).cuda()
    d_bits = torch.zeros((n_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_corr = torch.zeros((n_frames, 76, 2), dtype=torch.float32, device="cuda")
    d_fft = torch.zeros((n_frames, 77, 2048, 2), dtype=torch.float32, device="cuda")
    ctx.ofdm_demod_frames_raw(d_raw, fmt, n_frames, d_bits, freq_offset=d_freq, cp_corr=d_corr, fft=d_fft)
    torch.cuda.synchronize()
    # the product's two-step path
    d_iq = torch.empty(n_frames * dabgpu.NB_FRAME_SAMPLES * 2, dtype=torch.float32, device="cuda")
    d_bits2 = torch.zeros_like(d_bits)
    ctx.iq_convert(d_raw, fmt, n_frames * dabgpu.NB_FRAME_SAMPLES, d_iq)
    ctx.ofdm_demod_frames(d_iq, d_bits2, freq_offset=d_freq, n_frames=n_frames)
    torch.cuda.synchronize()
    assert torch.equal(d_bits, d_bits2)
    iq = oracle.iq_convert(raw, fmt).view(np.complex64).reshape(n_frames, -1)
    m = oracle.mapper()
    for k in range(n_frames):
        r = oracle.demod_frame(iq[k], float(freq[k]), want_fft=True, m=m)
        assert np.array_equal(d_bits[k].cpu().numpy(), r["bits"]), (name, k)
        assert np.array_equal(d_corr[k].cpu().numpy().view(np.uint32).reshape(-1), r["cp_corr"].view(np.uint32).reshape(-1)), (name, k)
        assert np.array_equal(d_fft[k].cpu().numpy().view(np.uint32).reshape(-1), r["fft"].view(np.uint32).reshape(-1)), (name, k)
