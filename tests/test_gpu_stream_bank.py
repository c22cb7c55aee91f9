"""-m gpu parity test of the device-side unsynchronised front end (dabgpu_stream_bank_*): several
receivers with different carrier offsets, timing, noise and signal drop-outs advance in one bank; every completed frame's
soft bits and every status field after every call must equal tests/stream_model.py (the oracle-composed OFDM_Demod state
machine) fed the same blocks -- floats as uint32 bit patterns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_stream(oracle, seed, n_frames, cfo, pad, noise, dropout=None, amplitude=1.0 / 39.2):
    rng = np.random.default_rng(seed)
    frames = [oracle.modulate_frame(rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)) for _ in range(n_frames)]
    tx = oracle.apply_pll(np.concatenate(frames), cfo, 0.11)
    stream = np.concatenate([tx[oracle.NB_NULL_PERIOD:oracle.NB_NULL_PERIOD + 20000 + pad], tx])
    if dropout is not None:                    # the signal vanishes for a while: desync, NULL search, re-acquisition
        a, b = dropout
        stream[a:b] = 0
    stream = stream + noise * (rng.standard_normal(stream.size) + 1j * rng.standard_normal(stream.size))
    return (stream * amplitude).astype(np.complex64)


def noise_with_dips(seed, streams_len):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(streams_len) + 1j * rng.standard_normal(streams_len)) * 0.1
    for a in range(40000, streams_len - 4000, 53000):
        x[a:a + 2600] *= 0.05
    return x.astype(np.complex64)


@pytest.mark.parametrize("block", [65536, 10007, 500000])
def test_stream_bank_matches_stream_model(oracle, block):
    import dabgpu
    import stream_model as SM
    import torch
    ctx = dabgpu.Context(0)
    streams = [
        make_stream(oracle, 1, 4, 1.8e-3, 1234, 3.0),
        make_stream(oracle, 2, 4, -7.3e-3, 77, 6.0),
        make_stream(oracle, 3, 4, 2.0e-4, 2551, 1.0, dropout=(215000, 235000)),        # wipes the PRS of the 2nd frame
        noise_with_dips(4, streams_len=850000),                                    # no DAB signal: false NULLs, failed syncs
    ]
    n = min(s.size for s in streams)
    n -= n % 7                                                                     # a ragged last block
    streams = [s[:n] for s in streams]
    E = len(streams)
    models = [SM.StreamModel(oracle) for _ in range(E)]
    bank = dabgpu.StreamBank(ctx, E)
    max_frames = block // 191400 + 2
    d_bits = torch.zeros((E, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_nf = torch.zeros(E, dtype=torch.int32, device="cuda")
    total = [0] * E
    for k in range(0, n, block):
        m = min(block, n - k)
        blk = np.stack([s[k:k + m] for s in streams])
        d_iq = torch.view_as_real(torch.from_numpy(blk).cuda())
        d_bits.zero_()
        bank.process(d_iq, m, m, d_bits, max_frames, d_nf)
        torch.cuda.synchronize()
        nf = d_nf.cpu().numpy()
        st = bank.status()
        for e in range(E):
            before = len(models[e].out_frames)
            models[e].process(blk[e])
            new = models[e].out_frames[before:]
            assert nf[e] == len(new), (block, k, e, nf[e], len(new))
            for j, fr in enumerate(new):
                assert np.array_equal(d_bits[e, j].cpu().numpy(), fr["bits"]), (block, k, e, j)
            total[e] += len(new)
            mo = models[e]
            assert int(st["state"][e]) == mo.state, (block, k, e)
            assert st["signal_l1_average"][e].view(np.uint32) == np.float32(mo.signal_avg).view(np.uint32), (block, k, e)
            assert st["freq_coarse"][e].view(np.uint32) == np.float32(mo.sync.freq_coarse).view(np.uint32), (block, k, e)
            assert st["freq_fine"][e].view(np.uint32) == np.float32(mo.sync.freq_fine).view(np.uint32), (block, k, e)
            assert int(st["fine_time_offset"][e]) == mo.fine_time_offset
            assert int(st["total_frames_read"][e]) == mo.frames_read and int(st["total_frames_desync"][e]) == mo.frames_desync
    assert total[0] >= 3 and total[1] >= 3 and total[2] >= 1, total
    assert models[2].frames_desync >= 1, "the drop-out must force a re-acquisition"
    assert models[3].frames_desync >= 3 and total[3] == 0, "every false NULL must end in a failed impulse-peak test"
    bank.close()


def test_stream_bank_reset_and_argument_checks(oracle):
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    bank = dabgpu.StreamBank(ctx, 2)
    s = make_stream(oracle, 9, 2, 1e-3, 10, 2.0)[:300000]
    d_iq = torch.view_as_real(torch.from_numpy(np.stack([s, s])).cuda())
    d_bits = torch.zeros((2, 3, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_nf = torch.zeros(2, dtype=torch.int32, device="cuda")
    bank.process(d_iq, s.size, s.size, d_bits, 3, d_nf)
    first = (d_bits.cpu().numpy().copy(), d_nf.cpu().numpy().copy(), bank.status().copy())
    assert first[1][0] == 1 and first[1][1] == 1 and np.array_equal(first[0][0], first[0][1])
    bank.reset()
    d_bits.zero_()
    bank.process(d_iq, s.size, s.size, d_bits, 3, d_nf)
    assert np.array_equal(d_bits.cpu().numpy(), first[0]) and np.array_equal(bank.status(), first[2])
    with pytest.raises(dabgpu.DabGpuError):
        bank.process(d_iq, s.size, s.size, d_bits, 1, d_nf)          # too few output slots for this block length
    bank.close()


@pytest.mark.parametrize("name", ["raw_u8", "raw_s8", "raw_s16l", "raw_u16l"])
def test_stream_bank_from_capture_formats(oracle, name):
    """SDR-style quantised streams: u8 / s8 / s16l blocks are read by the bank's kernels themselves (no conversion pass),
    other formats are dequantised into bank scratch first; frames equal the oracle state machine fed oracle.iq_convert(raw)
    in the same blocks"""
    import dabgpu
    import stream_model as SM
    import torch
    ctx = dabgpu.Context(0)
    fmt = dabgpu.IQ_FORMATS.index(name)
    # fused formats: any block boundary (+2 samples: odd byte alignments inside the raw data); converted formats: 16-byte aligned blocks
    E, block = 2, (65536 + 2 if name != "raw_u16l" else 65536)
    streams = [make_stream(oracle, 21, 3, 1.1e-3, 500, 2.0), make_stream(oracle, 22, 3, -0.7e-3, 1700, 4.0)]
    n = min(s.size for s in streams) // 8 * 8
    raws = []
    for s in streams:
        x = np.stack([s[:n].real, s[:n].imag], axis=-1).reshape(-1)
        x = x / np.abs(x).max()
        if name == "raw_u8":
            raws.append(np.clip(np.rint(x * 127.0 + 127.5), 0, 255).astype(np.uint8))
        elif name == "raw_s8":
            raws.append(np.clip(np.rint(x * 127.0), -128, 127).astype(np.int8).view(np.uint8))
        elif name == "raw_s16l":
            raws.append(np.clip(np.rint(x * 30000.0), -32768, 32767).astype("<i2").view(np.uint8))
        else:
            raws.append((np.clip(np.rint(x * 30000.0), -32768, 32767).astype(np.int32) + 32768).astype("<u2").view(np.uint8))
    raw = np.stack(raws)                                                    # [E][n * sample bytes]
    sb = dabgpu.iq_format_sample_bytes(fmt)
    d_raw = torch.from_numpy(raw).cuda()
    bank = dabgpu.StreamBank(ctx, E)
    max_frames = block // 191400 + 2
    d_bits = torch.zeros((E, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_nf = torch.zeros(E, dtype=torch.int32, device="cuda")
    models = [SM.StreamModel(oracle) for _ in range(E)]
    iq = [oracle.iq_convert(raw[e], fmt).view(np.complex64) for e in range(E)]
    got = 0
    for k in range(0, n, block):
        m = min(block, n - k)
        bank.process_raw(d_raw[:, sb * k:].data_ptr(), fmt, n, m, d_bits, max_frames, d_nf)
        torch.cuda.synchronize()
        nf = d_nf.cpu().numpy()
        for e in range(E):
            before = len(models[e].out_frames)
            models[e].process(iq[e][k:k + m])
            new = models[e].out_frames[before:]
            assert nf[e] == len(new), (name, k, e)
            for j, fr in enumerate(new):
                assert np.array_equal(d_bits[e, j].cpu().numpy(), fr["bits"]), (name, k, e, j)
            got += len(new)
    st = bank.status()
    for e in range(E):
        assert st["freq_fine"][e].view(np.uint32) == np.float32(models[e].sync.freq_fine).view(np.uint32)
        assert st["signal_l1_average"][e].view(np.uint32) == np.float32(models[e].signal_avg).view(np.uint32)
    assert got >= 3
    bank.close()


@pytest.mark.parametrize("mode", [2, 3, 4])
@pytest.mark.parametrize("fmt", ["c32", "raw_s16l"])
def test_stream_bank_in_other_transmission_modes(oracle, mode, fmt):
    """dabgpu_stream_bank_create_mode: three receivers of transmission mode II / III / IV (different offsets, one with a signal
    drop-out) vs the oracle state machine of that mode (tests/stream_model.py), blocks in complex float or raw s16 capture
    bytes -- every frame's soft bits and every status field after every call, floats as bit patterns"""
    import dabgpu
    import modes_model as MM
    import stream_model as SM
    import torch
    ctx = dabgpu.Context(0)
    g = oracle.geometry(mode)

    def make(seed, n_frames, cfo_bins, pad, noise, dropout=None):
        rng = np.random.default_rng(1000 * mode + seed)
        sent = [rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8) for _ in range(n_frames)]
        tx = oracle.apply_pll(np.concatenate([MM.make_tx_frame(oracle, mode, b, rng) for b in sent]), cfo_bins / g.nb_fft, 0.2)
        s = np.concatenate([tx[g.nb_null_period:g.nb_null_period + 6000 + pad], tx])
        if dropout is not None:
            s[dropout[0]:dropout[1]] = 0
        s = (s + noise * (rng.standard_normal(s.size) + 1j * rng.standard_normal(s.size))) / 39.2
        return s.astype(np.complex64)

    fs = g.nb_frame_samples
    streams = [make(1, 7, 2.1, 77, 0.05), make(2, 7, -3.4, 311, 0.1), make(3, 7, 0.3, 5, 0.05, dropout=(2 * fs + 9000, 2 * fs + 9000 + fs // 3))]
    n = min(s.size for s in streams)
    n -= n % 5
    streams = [s[:n] for s in streams]
    fnum = dabgpu.IQ_FORMATS.index("raw_s16l")
    if fmt == "raw_s16l":                                # quantise first: model and bank must see the same samples
        q = [np.clip(np.rint(s.view(np.float32) * 32767.0 * 8.0), -32768, 32767).astype("<i2") for s in streams]
        streams = [oracle.iq_convert(x.view(np.uint8), fnum).view(np.complex64) for x in q]
    E = len(streams)
    cfg = dabgpu.StreamCfg()
    dabgpu.lib().dabgpu_stream_cfg_default(dabgpu.C.byref(cfg))
    cfg.sync.impulse_peak_threshold_db = 8.0             # the reference's default 20 dB rarely passes with the short symbols of modes II / III
    bank = dabgpu.StreamBank(ctx, E, cfg, mode=mode)
    models = [SM.StreamModel(oracle, mode) for _ in range(E)]
    for m in models:
        m.cfg.impulse_peak_threshold_db = 8.0
    block = 40000
    max_frames = block // (fs - g.nb_null_period - g.nb_symbol_period) + 2
    d_bits = torch.zeros((E, max_frames, g.nb_frame_bits), dtype=torch.int8, device="cuda")
    d_nf = torch.zeros(E, dtype=torch.int32, device="cuda")
    total = [0] * E
    for k in range(0, n, block):
        m = min(block, n - k)
        blk = np.stack([s[k:k + m] for s in streams])
        d_bits.zero_()
        if fmt == "c32":
            bank.process(torch.view_as_real(torch.from_numpy(blk).cuda()), m, m, d_bits, max_frames, d_nf)
        else:
            raw = np.stack([x[2 * k:2 * (k + m)] for x in q])
            bank.process_raw(torch.from_numpy(raw.view(np.uint8)).cuda(), fnum, m, m, d_bits, max_frames, d_nf)
        torch.cuda.synchronize()
        nf = d_nf.cpu().numpy()
        st = bank.status()
        for e in range(E):
            before = len(models[e].out_frames)
            models[e].process(blk[e])
            new = models[e].out_frames[before:]
            assert nf[e] == len(new), (mode, k, e, nf[e], len(new))
            for j, fr in enumerate(new):
                assert np.array_equal(d_bits[e, j].cpu().numpy(), fr["bits"]), (mode, k, e, j)
            total[e] += len(new)
            mo = models[e]
            assert int(st["state"][e]) == mo.state, (mode, k, e)
            assert st["signal_l1_average"][e].view(np.uint32) == np.float32(mo.signal_avg).view(np.uint32), (mode, k, e)
            assert st["freq_coarse"][e].view(np.uint32) == np.float32(mo.sync.freq_coarse).view(np.uint32), (mode, k, e)
            assert st["freq_fine"][e].view(np.uint32) == np.float32(mo.sync.freq_fine).view(np.uint32), (mode, k, e)
            assert int(st["fine_time_offset"][e]) == mo.fine_time_offset
            assert int(st["total_frames_read"][e]) == mo.frames_read and int(st["total_frames_desync"][e]) == mo.frames_desync
    assert sum(total) >= 6, total
    bank.close()


def test_clean_streams_starting_anywhere_count_the_desyncs_the_model_counts(oracle):
    """tools/bench_stream.py and tools/bench_ingest.py report a few `desync`s on CLEAN synthetic streams.  They are the reference
    algorithm's own start-up behaviour, not a defect of the bank: a receiver switched on in the middle of a NULL symbol (or whose first
    level estimate is still rising) takes a false NULL end, fails the impulse-peak test (ofdm_demodulator.cpp:525-529: Reset(),
    total_frames_desync++) and locks on the next NULL.  Streams that start at every kind of position must give exactly the model's
    counters."""
    import dabgpu
    import stream_model as SM
    import torch
    ctx = dabgpu.Context(0)
    rng = np.random.default_rng(9)
    frames = [oracle.modulate_frame(rng.integers(0, 2, oracle.NB_FRAME_BITS, dtype=np.uint8)) for _ in range(4)]
    tx = oracle.apply_pll(np.concatenate(frames), 1.1e-3, 0.0)
    starts = [0, 900, 1300, 2000, 2600, 2656 + 1000, 2656 + 2552 + 17, 100000, 196608 - 1500, 196608 - 300]
    n = 3 * 196608
    streams = [((tx[s:s + n] + 0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))) / 39.2).astype(np.complex64) for s in starts]
    E = len(streams)
    bank = dabgpu.StreamBank(ctx, E)
    block = 196608
    max_frames = 3
    d_bits = torch.zeros((E, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_nf = torch.zeros(E, dtype=torch.int32, device="cuda")
    models = [SM.StreamModel(oracle) for _ in range(E)]
    for k in range(0, n, block):
        blk = np.stack([s[k:k + block] for s in streams])
        bank.process(torch.view_as_real(torch.from_numpy(blk).cuda()), block, block, d_bits, max_frames, d_nf)
        torch.cuda.synchronize()
        for e in range(E):
            models[e].process(blk[e])
    st = bank.status()
    for e in range(E):
        assert int(st["total_frames_desync"][e]) == models[e].frames_desync, (starts[e], int(st["total_frames_desync"][e]), models[e].frames_desync)
        assert int(st["total_frames_read"][e]) == models[e].frames_read
    assert sum(m.frames_desync for m in models) >= 1, "expected at least one start position with a false first NULL"
    assert all(m.frames_read >= 1 for m in models), "every receiver locks within three frames"
    bank.close()


def test_bank_of_1024_streams_in_two_lanes_equals_the_small_bank(oracle):
    """from 1024 streams on a mode I bank runs the two halves of its streams on two HIP streams: 1024 receivers fed four distinct
    captures (each 256 times, interleaved so that every capture is in both halves) must give, stream by stream, the frames and the
    status of a 4-stream bank fed the same blocks -- bits, frame counts and every status field after every call"""
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    base = [make_stream(oracle, 11, 3, 1.8e-3, 1234, 3.0), make_stream(oracle, 12, 3, -7.3e-3, 77, 6.0),
            make_stream(oracle, 13, 3, 2.0e-4, 2551, 1.0, dropout=(215000, 235000)), noise_with_dips(14, streams_len=650000)]
    n = min(s.size for s in base)
    n -= n % 2
    q = [np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 127.0 + 127.5), 0, 255).astype(np.uint8)
         for s in base]                                                      # raw_u8 captures: 2 bytes per sample
    small = np.stack(q)                                                      # [4][n][2]
    N = 1024
    d_small = torch.from_numpy(small).cuda()
    d_big = d_small.repeat(N // 4, 1, 1).contiguous()                        # stream s = capture s mod 4
    fmt = dabgpu.IQ_FORMATS.index("raw_u8")
    block, max_frames = 300000, 3
    banks = {4: dabgpu.StreamBank(ctx, 4), N: dabgpu.StreamBank(ctx, N)}
    bits = {4: torch.zeros((4, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda"),
            N: torch.zeros((N, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")}
    nf = {4: torch.zeros(4, dtype=torch.int32, device="cuda"), N: torch.zeros(N, dtype=torch.int32, device="cuda")}
    frames = 0
    for k in range(0, n, block):
        m = min(block, n - k)
        for E, d in ((4, d_small), (N, d_big)):
            bits[E].zero_()
            banks[E].process_raw(d[:, k:].data_ptr(), fmt, n, m, bits[E], max_frames, nf[E])
        torch.cuda.synchronize()
        n4, nN = nf[4].cpu().numpy(), nf[N].cpu().numpy()
        assert np.array_equal(nN, np.tile(n4, N // 4)), k
        assert torch.equal(bits[N].view(N // 4, 4, -1), bits[4].view(1, 4, -1).expand(N // 4, -1, -1)), k
        s4, sN = banks[4].status(), banks[N].status()
        for name in s4.dtype.names:
            assert np.array_equal(sN[name].view(np.uint32) if sN[name].dtype == np.float32 else sN[name],
                                  np.tile(s4[name].view(np.uint32) if s4[name].dtype == np.float32 else s4[name], N // 4)), (k, name)
        frames += int(n4.sum())
    assert frames >= 5


@pytest.mark.parametrize("block", [300000, 196608, 500001, 65536])
@pytest.mark.parametrize("name", ["raw_f32l", "raw_u8", "raw_s16l"])
def test_retained_blocks_equal_the_copying_bank(oracle, name, block):
    """dabgpu_stream_bank_process_retained (the caller keeps a block valid until the next call has returned; no carry-over copy: the
    next call's demodulator reads a frame's head from the previous block) against dabgpu_stream_bank_process_raw on the same blocks:
    frame counts, every frame's soft bits and every status field after every call, for blocks longer than, equal to and shorter than
    a frame (the short ones copy the carried samples out of the previous block first), odd block lengths (pair boundaries), drop-outs
    (re-acquisition with a frame under collection) and a noise-only stream."""
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    base = [make_stream(oracle, 21, 4, 1.8e-3, 1235, 3.0), make_stream(oracle, 22, 4, -7.3e-3, 78, 6.0),
            make_stream(oracle, 23, 4, 2.0e-4, 2551, 1.0, dropout=(215000, 235000)), noise_with_dips(24, streams_len=850000)]
    n = min(s.size for s in base)
    E = len(base)
    fmt = dabgpu.IQ_FORMATS.index(name)
    if name == "raw_f32l":
        q = np.stack([np.stack([s[:n].real, s[:n].imag], axis=-1).astype(np.float32) for s in base])
    elif name == "raw_u8":
        q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 127.0 + 127.5), 0, 255).astype(np.uint8) for s in base])
    else:
        q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 30000.0), -32768, 32767).astype(np.int16) for s in base])
    max_frames = block // 191400 + 2
    banks = [dabgpu.StreamBank(ctx, E), dabgpu.StreamBank(ctx, E)]                 # [0] copies, [1] retained
    bufs = [torch.zeros((E, block, 2), dtype=torch.from_numpy(q[:1, :1]).dtype, device="cuda") for _ in range(3)]
    bits = [torch.zeros((E, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda") for _ in range(2)]
    nf = [torch.zeros(E, dtype=torch.int32, device="cuda") for _ in range(2)]
    prev = None
    frames, call = 0, 0
    for k in range(0, n, block):
        m = min(block, n - k)
        cur = bufs[1 + call % 2]                                                 # the retained bank's two alternating buffers
        cur[:, :m] = torch.from_numpy(q[:, k:k + m]).cuda()
        bufs[0][:, :m] = cur[:, :m]
        for b in bits:
            b.zero_()
        banks[0].process_raw(bufs[0], fmt, block, m, bits[0], max_frames, nf[0])
        banks[1].process_retained(cur, fmt, block, m, prev, bits[1], max_frames, nf[1])
        torch.cuda.synchronize()
        bufs[0].fill_(0 if name != "raw_u8" else 7)                              # the copying bank's block is free to go
        assert torch.equal(nf[0], nf[1]), (k, nf[0].tolist(), nf[1].tolist())
        assert torch.equal(bits[0], bits[1]), k
        s0, s1 = banks[0].status(), banks[1].status()
        for f in s0.dtype.names:
            a, b = (x[f].view(np.uint32) if x[f].dtype == np.float32 else x[f] for x in (s0, s1))
            assert np.array_equal(a, b), (k, f)
        frames += int(nf[0].sum().item())
        prev, call = cur, call + 1
    assert frames >= 6
    # a plain call while a block is retained is refused; after a release the bank goes on as a copying bank, identically
    with pytest.raises(RuntimeError):
        banks[1].process_raw(bufs[0], fmt, block, 1000, bits[1], max_frames, nf[1])
    banks[1].release(prev, fmt, block)
    tail = torch.from_numpy(q[:, :min(block, 250000)]).cuda()
    for bk, bt, cnt in ((banks[0], bits[0], nf[0]), (banks[1], bits[1], nf[1])):
        bufs[0][:, :tail.shape[1]] = tail
        bt.zero_()
        bk.process_raw(bufs[0], fmt, block, tail.shape[1], bt, max_frames, cnt)
    torch.cuda.synchronize()
    assert torch.equal(nf[0], nf[1]) and torch.equal(bits[0], bits[1])
    for bk in banks:
        bk.close()


def test_retained_blocks_in_two_lanes(oracle):
    """1024 streams (two lanes), raw_u8, 4-frame-sized blocks: the retained bank equals the copying bank stream by stream"""
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    base = [make_stream(oracle, 31, 6, 1.8e-3, 1234, 3.0), make_stream(oracle, 32, 6, -7.3e-3, 77, 6.0),
            make_stream(oracle, 33, 6, 2.0e-4, 2551, 1.0, dropout=(415000, 435000)), make_stream(oracle, 34, 6, 5.0e-3, 100001, 2.0)]
    n = min(s.size for s in base)
    q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 127.0 + 127.5), 0, 255).astype(np.uint8) for s in base])
    N, block, max_frames = 1024, 400001, 4
    fmt = dabgpu.IQ_FORMATS.index("raw_u8")
    banks = [dabgpu.StreamBank(ctx, N), dabgpu.StreamBank(ctx, N)]
    bufs = [torch.zeros((N, block, 2), dtype=torch.uint8, device="cuda") for _ in range(3)]
    bits = [torch.zeros((N, max_frames, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda") for _ in range(2)]
    nf = [torch.zeros(N, dtype=torch.int32, device="cuda") for _ in range(2)]
    prev, call, frames = None, 0, 0
    for k in range(0, n, block):
        m = min(block, n - k)
        cur = bufs[1 + call % 2]
        cur[:, :m] = torch.from_numpy(q[:, k:k + m]).cuda().repeat(N // 4, 1, 1)
        bufs[0][:, :m] = cur[:, :m]
        for b in bits:
            b.zero_()
        banks[0].process_raw(bufs[0], fmt, block, m, bits[0], max_frames, nf[0])
        banks[1].process_retained(cur, fmt, block, m, prev, bits[1], max_frames, nf[1])
        torch.cuda.synchronize()
        assert torch.equal(nf[0], nf[1]) and torch.equal(bits[0], bits[1]), k
        s0, s1 = banks[0].status(), banks[1].status()
        for f in s0.dtype.names:
            a, b = (x[f].view(np.uint32) if x[f].dtype == np.float32 else x[f] for x in (s0, s1))
            assert np.array_equal(a, b), (k, f)
        frames += int(nf[0][:4].sum().item())
        prev, call = cur, call + 1
    assert frames >= 10
    for bk in banks:
        bk.close()


@pytest.mark.parametrize("block,layout", [(191400, 1), (191399, 0), (150001, 1), (60000, 0)])
@pytest.mark.parametrize("name", ["raw_u8", "raw_f32l", "raw_s16l"])
def test_ring_form_with_retained_blocks_equals_the_copying_ring_form(oracle, name, block, layout):
    """dabgpu_stream_bank_process_ring_retained against dabgpu_stream_bank_process_ring_layout on the same blocks: the ring form's blocks
    are shorter than a frame, so a frame regularly lies in three places -- frame buffer (the PRS head), previous block, current block --
    and a frame that began in the last samples of a block cannot complete in the next one (those samples are copied at the start of the
    next call).  Ring contents, slots and every status field after every call; the largest block the ring form takes, an odd one (pair
    boundaries), a shorter one and one that takes the bulk-copy path; drop-outs (re-acquisition) and a noise-only stream."""
    import dabgpu
    import torch
    ctx = dabgpu.Context(0)
    base = [make_stream(oracle, 41, 5, 1.8e-3, 1235, 3.0), make_stream(oracle, 42, 5, -7.3e-3, 78, 6.0),
            make_stream(oracle, 43, 5, 2.0e-4, 191400 - 2656 - 20000 - 3000, 1.0),      # its first frame begins in the last 3000 samples of a 191400-sample block
            make_stream(oracle, 44, 5, 3.1e-3, 2551, 1.0, dropout=(415000, 435000)), noise_with_dips(45, streams_len=1000000)]
    n = min(s.size for s in base)
    E, H = len(base), 6
    fmt = dabgpu.IQ_FORMATS.index(name)
    if name == "raw_f32l":
        q = np.stack([np.stack([s[:n].real, s[:n].imag], axis=-1).astype(np.float32) for s in base])
    elif name == "raw_u8":
        q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 127.0 + 127.5), 0, 255).astype(np.uint8) for s in base])
    else:
        q = np.stack([np.clip(np.rint(np.stack([s[:n].real, s[:n].imag], axis=-1) / np.abs(s[:n]).max() * 30000.0), -32768, 32767).astype(np.int16) for s in base])
    banks = [dabgpu.StreamBank(ctx, E), dabgpu.StreamBank(ctx, E)]                 # [0] copies, [1] retained
    bufs = [torch.zeros((E, block, 2), dtype=torch.from_numpy(q[:1, :1]).dtype, device="cuda") for _ in range(3)]
    hist = [torch.zeros((E, H, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda") for _ in range(2)]
    slot = [torch.full((E,), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
    prev, call, frames = None, 0, 0
    for k in range(0, n, block):
        m = min(block, n - k)
        cur = bufs[1 + call % 2]
        cur[:, :m] = torch.from_numpy(q[:, k:k + m]).cuda()
        bufs[0][:, :m] = cur[:, :m]
        banks[0].process_ring(bufs[0], fmt, block, m, hist[0], H, slot[0], bits_layout=layout)
        banks[1].process_ring_retained(cur, fmt, block, m, prev, hist[1], H, slot[1], bits_layout=layout)
        torch.cuda.synchronize()
        bufs[0].fill_(0 if name != "raw_u8" else 9)                              # the copying bank's block is free to go
        assert torch.equal(slot[0], slot[1]), (k, slot[0].tolist(), slot[1].tolist())
        assert torch.equal(hist[0], hist[1]), k
        s0, s1 = banks[0].status(), banks[1].status()
        for f in s0.dtype.names:
            a, b = (x[f].view(np.uint32) if x[f].dtype == np.float32 else x[f] for x in (s0, s1))
            assert np.array_equal(a, b), (k, f)
        frames += int((slot[0] >= 0).sum().item())
        prev, call = cur, call + 1
    assert frames >= 12 and hist[0].abs().sum().item() > 0
    for bk in banks:
        bk.close()
