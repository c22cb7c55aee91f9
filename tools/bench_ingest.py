#!/usr/bin/env python3
"""PCIe-inclusive throughput of the ingest path (never the bench.py `value`): capture bytes start in HOST memory, go through
the pinned ingest ring (dabgpu_ingest_*: copy stream, copy of batch k + 1 overlapped with the demodulation of batch k) and are
demodulated from their capture format.  Reports frames/s for
  * frame-aligned batches (dabgpu_ofdm_demod_frames_raw), raw_u8 and complex float,
  * the unsynchronised front end (dabgpu_stream_bank_process_raw) with 1 and 256 receivers fed raw_u8 blocks,
with and without the soft bits copied back to pinned host memory, next to the PCIe ceiling (bytes per frame / 63 GB/s).

    python tools/bench_ingest.py [--frames 256] [--batches 12]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, dabgpu, dabsynth

L = 196608


def aligned(ctx, fmt_name, frames, batches, depth, copy_back):
    dev = torch.device("cuda", 0)
    prs, mapper, _ = dabgpu.host_tables()
    iq, tx_bits, freq = dabsynth.random_frames(frames, 11, dev, mapper, prs)
    sv = torch.view_as_real(iq)
    if fmt_name == "raw_u8":
        peak = float(sv.abs().max().item())
        host = torch.clamp(torch.round(sv / peak * 127.0 + 127.5), 0, 255).to(torch.uint8).cpu().numpy().reshape(-1)
    else:
        host = sv.cpu().numpy().view(np.uint8).reshape(-1)
    fmt = dabgpu.IQ_FORMATS.index(fmt_name if fmt_name != "c32" else "raw_f32l")
    nbytes = host.size
    pipe = dabgpu.IngestPipe(ctx, nbytes, depth)
    bits = torch.empty((frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    h_bits = torch.empty((frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8).pin_memory() if copy_back else None
    for _ in range(depth):                       # the capture bytes sit in the pinned buffers (a reader would fread into them)
        pipe.acquire()[:] = host

    def batch():
        pipe.acquire()
        d = pipe.submit(nbytes)
        pipe.wait(d)
        ctx.ofdm_demod_frames_raw(d, fmt, frames, bits, freq_offset=freq)
        pipe.consumed(d)
        if copy_back:
            h_bits.copy_(bits, non_blocking=True)
    for _ in range(3):
        batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batches):
        batch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hard = (bits.view(frames, 75, 3072) >= 0).to(torch.uint8)
    ok = int((hard != tx_bits).sum().item()) == 0
    per_frame = nbytes / frames
    return {"path": "frame-aligned", "format": fmt_name, "frames_per_batch": frames, "depth": depth, "soft_bits_copied_back": copy_back,
            "frames_per_s": frames * batches / dt, "h2d_GBps": nbytes * batches / dt / 1e9, "pcie_ceiling_frames_per_s": 63e9 / per_frame,
            "hard_bits_equal_transmitted": ok}


def bank(ctx, streams, block_frames, calls, depth, retained=False):
    dev = torch.device("cuda", 0)
    prs, mapper, _ = dabgpu.host_tables()
    iq, _, _ = dabsynth.random_frames(min(streams, 32), 5, dev, mapper, prs)
    frame = torch.cat([iq[:, 76 * 2552:], iq[:, :76 * 2552]], dim=1)                   # NULL first: a continuous transmission
    n_block = block_frames * L
    total = n_block * (calls + 3) + L
    reps = total // L + 2
    sv = torch.view_as_real(frame)
    peak = float(sv.abs().max().item())
    fmt = dabgpu.IQ_FORMATS.index("raw_u8")
    g = torch.Generator(device=dev); g.manual_seed(1)
    shift = torch.randint(0, L, (streams,), generator=g, device=dev)
    blocks = []                                                                         # host blocks [call][stream][n_block][2] u8
    raw_all = torch.empty((streams, total, 2), dtype=torch.uint8, device=dev)
    for e in range(streams):
        x = sv[e % frame.shape[0]].repeat(reps, 1)[int(shift[e]):int(shift[e]) + total]
        raw_all[e] = torch.clamp(torch.round(x / peak * 127.0 + 127.5), 0, 255).to(torch.uint8)
    for k in range(calls + 3):
        blocks.append(raw_all[:, k * n_block:(k + 1) * n_block].contiguous().cpu().numpy().reshape(-1))
    del raw_all
    nbytes = blocks[0].size
    pipe = dabgpu.IngestPipe(ctx, nbytes, depth)
    sb = dabgpu.StreamBank(ctx, streams)
    max_frames = block_frames + 2
    bits = torch.zeros((streams, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
    nf = torch.zeros(streams, dtype=torch.int32, device=dev)
    got = []

    def submit(k):
        pipe.acquire()[:] = blocks[k]              # stands for the reader's fread into the pinned buffer (included in the timing)
        return pipe.submit(nbytes)
    d_next = submit(0)
    t0 = None
    d_prev = None                                  # retained blocks: the twin of the previous call stays untouched until this call is queued
    for k in range(calls + 3):
        if k == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter(); got = []
        d = d_next
        if k + 1 < calls + 3:
            d_next = submit(k + 1)               # the next block crosses PCIe while this one is processed
        pipe.wait(d)
        if retained:                               # (depth >= 3: one twin being filled, the current one, the previous one)
            sb.process_retained(d, fmt, n_block, n_block, d_prev, bits, max_frames, nf)
            if d_prev is not None:
                pipe.consumed(d_prev)
            d_prev = d
        else:
            sb.process_raw(d, fmt, n_block, n_block, bits, max_frames, nf)
            pipe.consumed(d)
        got.append(int(nf.sum().item()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if retained:
        sb.release(d_prev, fmt, n_block)
        pipe.consumed(d_prev)
    st = sb.status()
    return {"path": "stream bank (unsynchronised front end)" + (", retained blocks" if retained else ""), "format": "raw_u8", "streams": streams, "block_frames": block_frames, "depth": depth,
            "frames_per_s": sum(got) / dt, "x_realtime_per_stream": sum(got) / dt / streams / (2.048e6 / L), "h2d_GBps": nbytes * calls / dt / 1e9,
            "frames": sum(got), "desync": int(st["total_frames_desync"].sum()), "includes_host_fill_memcpy": True,
            "pcie_ceiling_frames_per_s": 63e9 / (2 * L)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--batches", type=int, default=12)
    a = ap.parse_args()
    ctx = dabgpu.Context(0)
    out = []
    for fmt in ("raw_u8", "c32"):
        for depth in (1, 3):
            out.append(aligned(ctx, fmt, a.frames, a.batches, depth, False))
    out.append(aligned(ctx, "raw_u8", a.frames, a.batches, 3, True))
    out.append(bank(ctx, 1, 4, 8, 2))
    out.append(bank(ctx, 256, 1, 8, 2))
    out.append(bank(ctx, 256, 1, 8, 3, retained=True))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
