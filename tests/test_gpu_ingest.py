"""-m gpu: the ingest pipe (dabgpu_ingest_*, SURVEY P2 / ofdm_demodulator.cpp:550-577, app_ofdm_blocks.h:45-58): capture bytes that start in
host memory, cross PCIe through the pinned ring and are demodulated from their capture format must give the bits of the device-resident
path and of the oracle; buffers are recycled correctly when batches are submitted ahead."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ingest_ring_feeds_the_raw_demodulator(oracle):
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    rng = np.random.default_rng(4)
    n_frames, n_batches, depth = 2, 5, 2
    fmt = dabgpu.IQ_FORMATS.index("raw_u8")
    pipe = dabgpu.IngestPipe(ctx, n_frames * 196608 * 2, depth)
    batches, expect = [], []
    for b in range(n_batches):
        raws, exps = [], []
        for f in range(n_frames):
            bits = rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)
            x = oracle.tx_to_frame_buffer(oracle.apply_pll(oracle.modulate_frame(bits), -2.0e-4, 0.0)) / 39.2
            q = np.clip(np.rint(np.stack([x.real, x.imag], -1) * 40.0 + 127.5), 0, 255).astype(np.uint8)
            raws.append(q.reshape(-1))
            comp = oracle.iq_convert(q.reshape(-1), fmt)                     # float32 components, the reference reader's arithmetic
            exps.append(oracle.demod_frame((comp[0::2] + 1j * comp[1::2]).astype(np.complex64), 2.0e-4)["bits"])
        batches.append(np.concatenate(raws))
        expect.append(np.stack(exps))
    freq = torch.full((n_frames,), 2.0e-4, dtype=torch.float32, device="cuda")
    outs = [torch.empty((n_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda") for _ in range(n_batches)]
    # batch k + 1 is copied ahead while batch k is demodulated
    pipe.acquire()[:] = batches[0]
    d_next = pipe.submit(batches[0].size)
    for k in range(n_batches):
        d = d_next
        if k + 1 < n_batches:
            pipe.acquire()[:] = batches[k + 1]
            d_next = pipe.submit(batches[k + 1].size)
        pipe.wait(d)
        ctx.ofdm_demod_frames_raw(d, fmt, n_frames, outs[k], freq_offset=freq)
        pipe.consumed(d)
    torch.cuda.synchronize()
    for k in range(n_batches):
        assert np.array_equal(outs[k].cpu().numpy(), expect[k]), k
    with pytest.raises(dabgpu.DabGpuError):
        pipe.wait(12345)                                         # not a buffer of this pipe
    pipe.close()
