"""-m gpu parity tests of the DAB+ outer-code kernel (dabgpu_dabplus_*) through the C ABI: fire-code
acquisition, 5-frame collection, RS(120,110), header walk, access-unit CRCs -- against the reference-generated golden
sequences, against the CPU oracle on a batch of differently sized / damaged / misaligned streams, and end to end behind
the MSC Viterbi kernel.  Integer work: every record field and every super-frame byte identical."""
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


def same_record(g, o):
    """device record g vs oracle record o (oracle has two extra leading fields)"""
    for k in ("rs_failed_index", "rs_corrected", "firecode_ok", "header_valid", "descriptor", "num_aus", "au_walk_stopped_at"):
        if int(g[k]) != int(o[k]):
            return False, k
    if int(g["au_crc_ok_mask"]) != int(o["au_crc_ok_mask"]):
        return False, "au_crc_ok_mask"
    if o["header_valid"] and not np.array_equal(g["au_start"], o["au_start"]):
        return False, "au_start"
    return True, ""


def test_golden_sequences_one_stream_host_path(ctx, oracle):
    import dabgpu
    from test_oracle_dabplus import check_against_reference_events
    gdp = np.load(os.path.join(ROOT, "tests", "golden", "dabplus_vectors.npz"))
    for name in [str(s) for s in gdp["seq_names"]]:
        frames, n = gdp[f"{name}_frames"], int(gdp[f"{name}_n"])
        bank = dabgpu.DabPlusBank(ctx, 1)
        for k in range(frames.shape[0]):
            done, wait, r, sf = bank.process_frame_host(frames[k][:n])
            rec = {"firecode_wait_failed": wait, "superframe_done": done}
            rec.update({f: r[f] for f in r.dtype.names})
            check_against_reference_events(rec, sf, gdp[f"{name}_ref"][k], gdp[f"{name}_au_len"][k], gdp[f"{name}_au_bytes"][k], (name, k))
        bank.close()


def test_batch_of_streams_matches_oracle(ctx, oracle):
    run_batch_of_streams(ctx, oracle, 5)


def run_batch_of_streams(ctx, oracle, seed, strict_mix=True):
    """(also driven with many seeds by tools/fuzz_dabplus.py)"""
    import dabgpu
    import dabplus_model as M
    import torch
    rng = np.random.default_rng(seed)
    sizes = [24, 48, 96, 120, 192, 264, 384, 576, 1536, 96, 192, 72]
    E, n_calls, F = len(sizes), 14, 7                  # 7 logical frames per call: super frames straddle calls
    total = n_calls * F
    max_n = max(sizes)
    streams = np.zeros((E, total + 12, max_n), np.uint8)
    for e, n in enumerate(sizes):
        fr = []
        n_rs = 5 * n // 120
        while len(fr) < total + 12:
            sf, _, _ = M.make_superframe(oracle, rng, n, dac_rate=int(rng.integers(0, 2)), sbr_flag=int(rng.integers(0, 2)),
                                         bad_au_crc=(0,) if rng.random() < 0.2 else ())
            mode = rng.integers(0, 6)
            errs = {0: {}, 1: {0: 1}, 2: {i: int(rng.integers(0, 6)) for i in range(n_rs)}, 3: {int(rng.integers(0, n_rs)): 6},
                    4: {i: 5 for i in range(n_rs)}, 5: {}}[int(mode)]
            sf = M.corrupt(rng, sf, errs)
            if mode == 5:
                sf[int(rng.integers(0, 11))] ^= 0x10   # header / fire code region damaged inside the RS capacity
            fr += list(sf.reshape(5, n))
        fr = fr[e % 5:]                                 # every stream starts at a different phase of its super frame
        for k in range(total):
            streams[e, k, :n] = fr[k]
    if E > 10:                                          # a burst of uncorrectable super frames: > 10 failures -> re-acquisition
        streams[10, 5:5 + 60] = rng.integers(0, 256, (60, max_n), dtype=np.uint8)[:, :]
    bank = dabgpu.DabPlusBank(ctx, E)
    d_off = torch.from_numpy((np.arange(E, dtype=np.uint64) * np.uint64((total + 12) * max_n)).astype(np.int64)).cuda()
    d_n = torch.from_numpy(np.array(sizes, np.int32)).cuda()
    max_sf = (F + 4) // 5
    d_sf = torch.zeros((E, max_sf, 5 * max_n), dtype=torch.uint8, device="cuda")
    rec_bytes = np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE).itemsize
    assert rec_bytes == 88
    d_res = torch.zeros((E, max_sf, rec_bytes), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((E, 4), dtype=torch.int32, device="cuda")
    d_frames = torch.from_numpy(streams).cuda()
    procs = [oracle.AacFrameProcessor() for _ in range(E)]
    seen = {"ok": 0, "rs_fail": 0, "fire_fail": 0, "wait": 0}
    for c in range(n_calls):
        view = d_frames[:, c * F:]
        bank.process(view.data_ptr(), d_off, max_n, d_n, F, d_sf, 5 * max_n, d_res, max_sf, d_cnt)
        torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)).reshape(E, max_sf)
        sfs, cnt = d_sf.cpu().numpy(), d_cnt.cpu().numpy()
        for e, n in enumerate(sizes):
            k_sf, waits = 0, 0
            for f in range(F):
                rc, o, sf_o = procs[e].process(streams[e, c * F + f, :n])
                waits += int(o["firecode_wait_failed"])
                if o["superframe_done"]:
                    g = res[e, k_sf]
                    ok, field = same_record(g, o)
                    assert ok, (c, e, f, field, g, o)
                    assert int(g["frame_index"]) == f
                    assert np.array_equal(sfs[e, k_sf, :5 * n], sf_o), (c, e, f)
                    seen["ok" if o["header_valid"] else ("rs_fail" if o["rs_failed_index"] >= 0 else "fire_fail")] += 1
                    k_sf += 1
            assert cnt[e, 0] == k_sf and cnt[e, 1] == waits, (c, e, cnt[e], k_sf, waits)
            seen["wait"] += waits
    if strict_mix:
        assert seen["ok"] > 20 and seen["rs_fail"] > 3 and seen["wait"] > 5, seen
    bank.close()
    return seen


def test_behind_the_msc_viterbi_kernel(ctx, oracle):
    """two DAB+ sub-channels: super frames -> convolutional code -> 16-CIF time interleaver -> soft bits -> MSC Viterbi kernel
    -> DAB+ kernel reading the decoder's output buffer in place; the access units that come out are the ones that went in"""
    import dabgpu
    import dabplus_model as M
    import torch
    rng = np.random.default_rng(8)
    subs = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(100, 24, eep_level=2, eep_type=0)]
    gsubs = [dabgpu.SubChannel(s.start_address, s.length, s.is_uep, s.uep_prot_index, s.eep_prot_level, s.eep_type) for s in subs]
    plans = [oracle.subchannel_plan(s) for s in subs]
    nbytes = [p[2] for p in plans]                     # 192 and 96 bytes per logical frame
    n_frames, H = 10, 5
    n_cif = 4 * n_frames
    truth = [[] for _ in subs]
    cifs = rng.integers(-127, 128, (n_cif, oracle.NB_CIF_BITS), dtype=np.int8)
    for si, (s, n) in enumerate(zip(subs, nbytes)):
        lf_bytes = []
        while len(lf_bytes) < n_cif:
            sf, aus, _ = M.make_superframe(oracle, rng, n)
            truth[si].append(aus)
            lf_bytes += list(M.corrupt(rng, sf, 2).reshape(5, n))          # 2 symbol errors in every RS codeword
        lf = np.stack([oracle.msc_encode_logical(s, b) for b in lf_bytes[:n_cif]])
        tx = oracle.time_interleave(lf)
        for t in range(n_cif):
            cifs[t, s.start_address * 64:(s.start_address + s.length) * 64] = oracle.soft_from_bits(tx[t])
    cif_out = sum(nbytes)
    hist = torch.zeros((1, H, oracle.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_out = torch.zeros((1, 4, cif_out), dtype=torch.uint8, device="cuda")
    d_res = torch.zeros((4 * len(subs), 16), dtype=torch.uint8, device="cuda")
    bank = dabgpu.DabPlusBank(ctx, len(subs))
    d_off = torch.tensor([0, nbytes[0]], dtype=torch.int64, device="cuda")
    d_n = torch.tensor(nbytes, dtype=torch.int32, device="cuda")
    d_sf = torch.zeros((len(subs), 1, 5 * max(nbytes)), dtype=torch.uint8, device="cuda")
    d_rec = torch.zeros((len(subs), 1, np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE).itemsize), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((len(subs), 4), dtype=torch.int32, device="cuda")
    got = [[] for _ in subs]
    for f in range(n_frames):
        frame = np.zeros((1, oracle.NB_FRAME_BITS), np.int8)
        frame[0, 9216:] = cifs[4 * f:4 * f + 4].reshape(-1)
        hist[:, f % H].copy_(torch.from_numpy(frame).cuda())
        ctx.msc_decode_frames(hist, 1, H * oracle.NB_FRAME_BITS, H, f % H, gsubs, d_out, 4 * cif_out, d_res)
        if f < 4:
            continue                                   # CIFs 0..14 give no logical frame yet (frame 3's last CIF gives the first)
        bank.process(d_out, d_off, cif_out, d_n, 4, d_sf, 5 * max(nbytes), d_rec, 1, d_cnt)
        torch.cuda.synchronize()
        rec = d_rec.cpu().numpy().view(np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)).reshape(len(subs))
        cnt, sfs = d_cnt.cpu().numpy(), d_sf.cpu().numpy()
        for si in range(len(subs)):
            if cnt[si, 0]:
                r = rec[si]
                assert r["header_valid"] and r["rs_failed_index"] < 0 and r["rs_corrected"] == 2 * (5 * nbytes[si] // 120)
                aus = [sfs[si, 0, r["au_start"][i]:r["au_start"][i + 1] - 2].copy() for i in range(r["num_aus"])]
                assert int(r["au_crc_ok_mask"]) == (1 << r["num_aus"]) - 1
                got[si].append(aus)
    for si in range(len(subs)):
        # (10 frames = 40 CIFs; a super frame whose fire code a channel error hits is skipped by the acquisition -- seen with DAB_FUZZ_OFFSET=12:
        # two instead of three super frames in the window; what matters is that every one that came out is the transmitted one, in order).
        # The committed seed delivers all three: a wrongly dropped super frame must not pass there
        assert len(got[si]) >= (2 if int(os.environ.get("DAB_FUZZ_OFFSET", "0") or 0) else 3)
        # logical frame t leaves the time de-interleaver at CIF t + 15; frames 0..3 were not fed, so the first super frame the
        # DAB+ stage can acquire starts at logical frame 5 (index 1) -- or a later one when a channel error sits in its fire code
        def same(a, b):
            return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
        j0 = next(j for j in range(len(truth[si])) if same(got[si][0], truth[si][j]))
        assert 1 <= j0 <= 3, (si, j0)
        for k, aus in enumerate(got[si]):
            assert same(aus, truth[si][j0 + k]), (si, k)
    bank.close()


def test_unsupported_frame_sizes_are_reported_per_stream(ctx):
    """logical frames above 1536 bytes / a super-frame slot below 5 frames: flagged in d_counts[s][1], never skipped in silence
    (include/dabgpu.h); a normal stream in the same launch is unaffected"""
    import dabgpu
    import torch
    n = 3
    bank = dabgpu.DabPlusBank(ctx, n)
    bank.reset()
    frame_bytes = torch.tensor([1600, 192, 400], dtype=torch.uint32, device="cuda")
    stride = 2048
    frames = torch.zeros((n, 5, stride), dtype=torch.uint8, device="cuda")
    offsets = torch.tensor([0, 5 * stride, 10 * stride], dtype=torch.uint64, device="cuda")
    sf_stride = 5 * 192                                              # enough for stream 1 only
    sf = torch.zeros((n, 1, sf_stride), dtype=torch.uint8, device="cuda")
    res = torch.zeros((n, 1, 96), dtype=torch.uint8, device="cuda")
    counts = torch.full((n, 4), 77, dtype=torch.int32, device="cuda")
    bank.process(frames, offsets, stride, frame_bytes, 5, sf, sf_stride, res, 1, counts)
    torch.cuda.synchronize()
    c = counts.cpu().numpy()
    assert c[0, 0] == 0 and c[0, 1] == -1                            # 1600-byte frames: not supported by the bank kernel
    assert c[2, 0] == 0 and c[2, 1] == -2                            # 5 x 400 bytes do not fit the 960-byte slot
    assert c[1, 1] >= 0 and c[1, 0] + c[1, 1] >= 1                  # the 192-byte stream was processed (a super frame attempted or frames dropped)


def test_frame_sizes_that_are_not_whole_codewords(ctx, oracle):
    """the reference accepts any logical frame of at least 11 bytes (aac_frame_processor.cpp:129-137): below 24 bytes a super frame holds
    NO RS codeword (nothing is corrected, the fire code and the header walk still run), sizes that are not multiples of 24 leave bytes
    outside the codewords -- records and super-frame bytes must equal the oracle's for all of them"""
    import dabgpu
    import torch
    rng = np.random.default_rng(17)
    sizes = [11, 12, 23, 24, 25, 47, 100]
    E, F = len(sizes), 10
    max_n = max(sizes)
    streams = np.zeros((E, F, max_n), np.uint8)
    for e, n in enumerate(sizes):
        for k in range(F // 5):
            sf = rng.integers(0, 256, 5 * n, dtype=np.uint8)
            fc = oracle.firecode_crc(sf[2:11])
            sf[0], sf[1] = fc >> 8, fc & 0xFF
            streams[e, 5 * k:5 * k + 5, :n] = sf.reshape(5, n)
    bank = dabgpu.DabPlusBank(ctx, E)
    d_off = torch.from_numpy((np.arange(E, dtype=np.int64) * (F * max_n))).cuda()
    d_n = torch.from_numpy(np.array(sizes, np.int32)).cuda()
    max_sf = 2
    d_sf = torch.zeros((E, max_sf, 5 * max_n), dtype=torch.uint8, device="cuda")
    rec = np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)
    d_res = torch.zeros((E, max_sf, rec.itemsize), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((E, 4), dtype=torch.int32, device="cuda")
    bank.process(torch.from_numpy(streams).cuda(), d_off, max_n, d_n, F, d_sf, 5 * max_n, d_res, max_sf, d_cnt)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(rec).reshape(E, max_sf)
    sfs, cnt = d_sf.cpu().numpy(), d_cnt.cpu().numpy()
    done = 0
    for e, n in enumerate(sizes):
        proc = oracle.AacFrameProcessor()
        k_sf = 0
        for f in range(F):
            _, o, sf_o = proc.process(streams[e, f, :n])
            if o["superframe_done"]:
                ok, field = same_record(res[e, k_sf], o)
                assert ok, (n, f, field, res[e, k_sf], o)
                assert np.array_equal(sfs[e, k_sf, :5 * n], sf_o), (n, f)
                k_sf += 1
        assert cnt[e, 0] == k_sf, (n, cnt[e], k_sf)
        done += k_sf
    assert done >= 2 * len(sizes) - 2
    bank.close()
