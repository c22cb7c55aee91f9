"""CPU test of the oracle's one-call receive chain (oracle/dab_oracle_chain.c, what bench.py's cpu_baseline_full times): it must be
exactly the per-frame composition of the oracle's own functions (the composition the -m gpu tests check the device against), and on
a clean coded stream it must return what was transmitted."""
import numpy as np

import stream_model as SM

P = 700


def test_receive_frames_equals_the_composed_oracle_and_the_transmitted_bytes(oracle):
    O = oracle
    subs = [O.subchannel(0, 48, eep_level=2, eep_type=0), O.subchannel(100, 24, eep_level=1, eep_type=0)]
    n_frames, cfo, toff = 7, 2.3e-3, 41
    stream, truth = SM.make_ensemble_stream(O, n_frames, subs, seed=12, cfo=cfo, timing_pad=0, noise=1.5, amplitude=1.0)
    lead = 30000                                                   # make_ensemble_stream's lead-in before the first NULL
    stride = P + 1544 + O.NB_FRAME_SAMPLES
    slices = np.zeros((n_frames, stride), np.complex64)
    for j in range(n_frames):
        prs_start = lead + O.NB_NULL_PERIOD + j * O.NB_FRAME_SAMPLES
        a = prs_start - (P + toff)                                  # slice sample P + toff = first PRS sample of frame j
        seg = stream[a:a + stride]
        slices[j, :seg.size] = seg
    # one C call over the 7 frames
    got = O.receive_frames(slices, stride, P, n_frames, subs)
    # (the first frame after acquisition is demodulated with the coarse estimate alone -- here 0.17 carrier spacings off, too much for
    # DQPSK -- exactly as in the reference; the fine loop has it from the second frame on)
    assert got["sync_failed"] == 0 and got["fib_crc_ok"] == 12 * (n_frames - 1)
    assert got["state"].total_frames_read == n_frames and got["state"].fine_time_offset == toff
    # the same, composed in Python from the per-function oracle
    conj_ref, time_ref = O.sync_refs()
    st = O.SyncState(0.0, 0.0, 0, 0, 0, 0)
    deint = [O.Deinterleaver(s.length * 8) for s in subs]
    for j in range(n_frames):
        prs_sym = slices[j, P:P + 2048]
        O.coarse_freq_sync(prs_sym, st, None, time_ref)
        f = np.float32(np.float32(st.freq_coarse) + np.float32(st.freq_fine))
        ok, off, _ = O.fine_time_sync(prs_sym, f, None, conj_ref)
        assert ok and off == toff
        r = O.demod_frame(slices[j, P + off:P + off + O.NB_FRAME_SAMPLES], f)
        st.freq_fine = float(O.update_fine_freq(st.freq_fine, r["total_phase"]))
        fib = np.stack([O.fic_decode_group(r["bits"][g * 2304:(g + 1) * 2304])[0] for g in range(4)])
        msc = []
        for c in range(4):
            row = []
            for si, s in enumerate(subs):
                cif = r["bits"][9216 + c * 55296:9216 + (c + 1) * 55296]
                deint[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                lf = deint[si].deinterleave()
                row.append(O.msc_decode_logical(s, lf)[0] if lf is not None else np.zeros(O.subchannel_plan(s)[2], np.uint8))
            msc.append(np.concatenate(row))
    assert np.float32(got["state"].freq_coarse).view(np.uint32) == np.float32(st.freq_coarse).view(np.uint32)
    assert np.float32(got["state"].freq_fine).view(np.uint32) == np.float32(st.freq_fine).view(np.uint32)
    assert np.array_equal(got["fib"], fib) and np.array_equal(got["msc"], np.stack(msc))
    # ... and what was transmitted: the last frame's FIBs, and logical frames 15 CIFs older than its CIFs (time interleaver)
    for g in range(4):
        for i in range(3):
            assert np.array_equal(got["fib"][g, 32 * i:32 * i + 30], truth["fibs"][n_frames - 1, g, 30 * i:30 * i + 30])
    off_b = 0
    for si, p in enumerate(truth["plans"]):
        for c in range(4):
            assert np.array_equal(got["msc"][c, off_b:off_b + p[2]], truth["payload"][si][4 * (n_frames - 1) + c - 15])
        off_b += p[2]


def test_receive_frames_resets_on_a_failed_synchronisation_and_checks_its_arguments(oracle):
    O = oracle
    rng = np.random.default_rng(3)
    stride = P + 1544 + O.NB_FRAME_SAMPLES
    noise = (rng.standard_normal((1, stride)) + 1j * rng.standard_normal((1, stride))).astype(np.complex64)
    got = O.receive_frames(noise, stride, P, 2, [])
    assert got["sync_failed"] == 2 and got["fib_crc_ok"] == 0 and got["state"].total_frames_desync == 2 and got["state"].total_frames_read == 0
    assert got["state"].freq_coarse == 0.0 and got["state"].is_found_coarse == 0
    import pytest
    with pytest.raises(ValueError):
        O.receive_frames(noise, stride, 100, 1, [])
