"""-m gpu end-to-end test of the device-resident pipeline: unsynchronised u8 capture streams of several ensembles ->
stream bank (frames into per-ensemble history rings) -> FIC + MSC ring decode -> DAB+ outer code, with no host work between
the stages except reading back what is compared.  Expected side: the CPU oracle composed the same way (StreamModel on the
oracle's dequantisation, fic_decode_group, Deinterleaver + msc_decode_logical, AacFrameProcessor).  Every FIB byte and CRC
mask, every decoded sub-channel byte and every super-frame record / byte must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", [0, 1], ids=["natural", "classed"])
def test_streams_to_access_units_on_the_device(oracle, layout):
    """layout = 1: the rings hold the MSC soft bits in time-interleaver class order (dabgpu_stream_bank_process_ring_layout /
    dabgpu_msc_decode_ring_layout): same FIBs, sub-channel bytes and access units; the ring content is the natural frame permuted"""
    import dabgpu
    import dabplus_model as M
    import stream_model as SM
    import torch
    rng = np.random.default_rng(99)
    ctx = dabgpu.Context(0)
    sub = oracle.subchannel(64, 48, eep_level=2, eep_type=0)                 # 192-byte logical frames: DAB+ at 64 kbit/s
    gsub = dabgpu.SubChannel(sub.start_address, sub.length, sub.is_uep, sub.uep_prot_index, sub.eep_prot_level, sub.eep_type)
    nbytes = oracle.subchannel_plan(sub)[2]
    E, n_frames, H, block = 3, 11, 6, 150000
    raws, sent_aus = [], []
    for e in range(E):
        lf, aus = [], []
        while len(lf) < 4 * n_frames + 5:
            sf, a, _ = M.make_superframe(oracle, rng, nbytes, dac_rate=e % 2, sbr_flag=1)
            aus.append(a)
            lf += list(M.corrupt(rng, sf, 1).reshape(5, nbytes))            # one symbol error in every RS codeword
        payload = [np.stack(lf[e:4 * n_frames + e])]                        # each ensemble at another super-frame phase
        stream, _ = SM.make_ensemble_stream(oracle, n_frames, [sub], seed=300 + e, cfo=(1.3e-3, -2.2e-3, 4e-4)[e],
                                            timing_pad=(77, 1999, 640)[e], noise=2.0, payload=payload)
        x = np.stack([stream.real, stream.imag], axis=-1).reshape(-1)
        raws.append(np.clip(np.rint(x / np.abs(x).max() * 127.0 + 127.5), 0, 255).astype(np.uint8))
        sent_aus.append(aus)
    n = min(r.size for r in raws) // 2
    raw = np.stack([r[:2 * n] for r in raws])
    d_raw = torch.from_numpy(raw).cuda()

    bank = dabgpu.StreamBank(ctx, E)
    dp = dabgpu.DabPlusBank(ctx, E)
    d_hist = torch.zeros((E, H, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device="cuda")
    d_slot = torch.full((E,), -1, dtype=torch.int32, device="cuda")
    d_fib = torch.zeros((E, 4, 96), dtype=torch.uint8, device="cuda")
    d_fres = torch.zeros((E * 4, 16), dtype=torch.uint8, device="cuda")
    d_msc = torch.zeros((E, 4, nbytes), dtype=torch.uint8, device="cuda")
    d_mres = torch.zeros((E * 4, 16), dtype=torch.uint8, device="cuda")
    rec_dt = np.dtype(dabgpu.SUPERFRAME_RESULT_DTYPE)
    d_sf = torch.zeros((E, 1, 5 * nbytes), dtype=torch.uint8, device="cuda")
    d_rec = torch.zeros((E, 1, rec_dt.itemsize), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((E, 4), dtype=torch.int32, device="cuda")
    d_off = (torch.arange(E, dtype=torch.int64, device="cuda") * (4 * nbytes))
    d_nb = torch.full((E,), nbytes, dtype=torch.int32, device="cuda")

    to_natural = dabgpu.classed_to_natural_index()
    models = [SM.StreamModel(oracle) for _ in range(E)]
    iq = [oracle.iq_convert(raw[e], 0).view(np.complex64) for e in range(E)]
    deint = [oracle.Deinterleaver(sub.length * 8) for _ in range(E)]
    aac = [oracle.AacFrameProcessor() for _ in range(E)]
    got_aus = [[] for _ in range(E)]
    checked = dict(frames=0, fibs=0, msc=0, superframes=0)
    for k in range(0, n, block):
        m = min(block, n - k)
        bank.process_ring(d_raw[:, 2 * k:].data_ptr(), 0, n, m, d_hist, H, d_slot, bits_layout=layout)
        ctx.fic_decode_ring(d_hist, E, H * dabgpu.NB_FRAME_BITS, d_slot, d_fib, d_fres)
        ctx.msc_decode_ring(d_hist, E, H * dabgpu.NB_FRAME_BITS, H, d_slot, [gsub], d_msc, 4 * nbytes, d_mres, bits_layout=layout)
        torch.cuda.synchronize()
        slot = d_slot.cpu().numpy()
        st = bank.status()
        # the DAB+ stage starts once all 4 CIFs of the newest frame come out of a full time de-interleaver (5th frame on)
        active = np.where((slot >= 0) & (st["total_frames_read"] >= 5), slot, -1).astype(np.int32)
        d_active = torch.from_numpy(active).cuda()
        dp.process_masked(d_msc, d_off, nbytes, d_nb, 4, d_sf, 5 * nbytes, d_rec, 1, d_cnt, d_active, 1)
        torch.cuda.synchronize()
        hist, fib, msc = d_hist.cpu().numpy(), d_fib.cpu().numpy(), d_msc.cpu().numpy()
        fres = d_fres.cpu().numpy().view(np.dtype(dabgpu.RESULT_DTYPE)).reshape(E, 4)
        rec = d_rec.cpu().numpy().view(rec_dt).reshape(E)
        cnt, sfs = d_cnt.cpu().numpy(), d_sf.cpu().numpy()
        for e in range(E):
            before = len(models[e].out_frames)
            models[e].process(iq[e][k:k + m])
            new = models[e].out_frames[before:]
            assert len(new) <= 1
            if not new:
                assert slot[e] == -1 and cnt[e, 0] == 0
                continue
            bits = new[0]["bits"]
            assert slot[e] == (models[e].frames_read - 1) % H
            assert np.array_equal(hist[e, slot[e]][to_natural] if layout else hist[e, slot[e]], bits), (k, e)
            checked["frames"] += 1
            for g in range(4):
                eb, em, ee = oracle.fic_decode_group(bits[g * 2304:(g + 1) * 2304], 0)
                assert np.array_equal(fib[e, g], eb) and int(fres[e, g]["crc_ok_mask"]) == em and int(fres[e, g]["path_error"]) == ee
                checked["fibs"] += bin(em).count("1")
            n_sf = 0
            for c in range(4):
                cif = bits[9216 + c * 55296:9216 + (c + 1) * 55296]
                deint[e].consume(cif[sub.start_address * 64:(sub.start_address + sub.length) * 64])
                lf = deint[e].deinterleave()
                if lf is None:
                    continue
                dec, _ = oracle.msc_decode_logical(sub, lf, 0)
                assert np.array_equal(msc[e, c], dec), (k, e, c)
                checked["msc"] += 1
                if active[e] < 0:
                    continue
                rc, o, sf_o = aac[e].process(dec)
                if o["superframe_done"]:
                    g = rec[e]
                    assert cnt[e, 0] == 1 and int(g["frame_index"]) == c
                    for f in ("rs_failed_index", "rs_corrected", "firecode_ok", "header_valid", "descriptor", "num_aus", "au_walk_stopped_at"):
                        assert int(g[f]) == int(o[f]), (k, e, c, f)
                    assert int(g["au_crc_ok_mask"]) == int(o["au_crc_ok_mask"]) and np.array_equal(sfs[e, 0], sf_o)
                    if o["header_valid"]:
                        got_aus[e].append([sf_o[o["au_start"][i]:o["au_start"][i + 1] - 2].copy() for i in range(o["num_aus"])])
                    checked["superframes"] += 1
                    n_sf += 1
            if active[e] >= 0:
                assert cnt[e, 0] == n_sf
    assert checked["frames"] >= E * (n_frames - 2) and checked["fibs"] >= 12 * E * (n_frames - 3) and checked["msc"] >= E * 4 * (n_frames - 6)
    assert checked["superframes"] >= 2 * E
    for e in range(E):                                                        # what came out is what was sent
        def same(a, b):
            return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
        assert len(got_aus[e]) >= 2
        j0 = next(j for j in range(len(sent_aus[e])) if same(got_aus[e][0], sent_aus[e][j]))
        assert all(same(a, sent_aus[e][j0 + i]) for i, a in enumerate(got_aus[e]))
    bank.close(); dp.close()
