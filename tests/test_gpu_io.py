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
