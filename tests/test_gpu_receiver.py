"""-m gpu: the receiver pipeline of the C ABI (dabgpu_receiver_*, include/dabgpu.h) driven directly through ctypes, against the oracle.
One receiver, seven steady-state frames with a carrier offset and a timing offset, every frame handed over the way the OFDM_Demod mirror
class does it -- "NULL | frame" assembled in the receiver's page-locked staging buffer, dabgpu_receiver_submit_sync as soon as the PRS slot
is there, the frame's end derived from the record, dabgpu_receiver_submit_frame, results collected LATER (three frames in flight) -- must
give what the oracle's per-frame composition gives (ofdm_demodulator.cpp:360-548 -> :650-766 -> :606-618; fic_decoder.cpp:53-117;
msc_decoder.cpp:46-154): soft bits byte for byte, the frequency words as float32 bit patterns, the fine time offset, FIB bytes + CRC masks and
the sub-channel bytes fetched from the receiver's frame session.  A failed impulse-peak test resets the device-resident state in stream
order (dabgpu_receiver_reset) and the next frame re-acquires exactly as the oracle does from a zeroed record."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Frame(C.Structure):
    _fields_ = [("generation", C.c_uint64), ("bits", C.c_void_p), ("n_bits", C.c_size_t), ("freq_fine", C.c_float), ("total_phase", C.c_float),
                ("fft", C.c_void_p), ("dqpsk", C.c_void_p)]


def _api(dabgpu):
    L = dabgpu.lib()
    L.dabgpu_receiver_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.dabgpu_receiver_destroy.argtypes = [C.c_void_p]
    L.dabgpu_receiver_session.restype = C.c_void_p
    L.dabgpu_receiver_session.argtypes = [C.c_void_p]
    L.dabgpu_receiver_set_subchannels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.dabgpu_receiver_stage.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.dabgpu_receiver_reset.argtypes = [C.c_void_p]
    L.dabgpu_receiver_submit_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.dabgpu_receiver_wait_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dabgpu_receiver_submit_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.dabgpu_receiver_wait_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.dabgpu_receiver_submit_demod.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.POINTER(C.c_uint64)]
    L.dabgpu_receiver_submit_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    L.dabgpu_frame_session_fetch_fib_group.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dabgpu_frame_session_fetch_cif.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("form", ["private", "banked"])
@pytest.mark.parametrize("calls", ["submit_frame", "submit_demod_then_decode"])
def test_receiver_pipeline_equals_the_oracle_composition(oracle, calls, form):
    """calls: dabgpu_receiver_submit_frame (one call: the decode waits for the demodulation on the device), or dabgpu_receiver_submit_demod followed -- here
    two frames LATER, as another thread would -- by dabgpu_receiver_submit_decode (the host waits; no stream waits for another): the same bytes.
    form: a receiver with its own pipeline, or a member of the device's receiver bank (dabgpu_receiver_create_banked: the calls post jobs, a bank
    thread issues them in rounds -- here rounds of one job): the same interface, the same bytes"""
    import dabgpu
    import stream_model as SM
    O = oracle
    L = _api(dabgpu)
    ck = lambda st, what: dabgpu.check(st, what)               # noqa: E731
    if form == "banked" and calls != "submit_frame":
        # a round of the bank decodes what it demodulates: the two-call form is refused, with a message
        L.dabgpu_receiver_create_banked.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        rx = C.c_void_p()
        ck(L.dabgpu_receiver_create_banked(C.byref(rx), 0), "dabgpu_receiver_create_banked")
        try:
            assert L.dabgpu_receiver_submit_demod(rx, 2656, 0.9, 0, None) == 5 and L.dabgpu_receiver_submit_decode(rx, 0, 0) == 5        # DABGPU_ERR_UNSUPPORTED
        finally:
            L.dabgpu_receiver_destroy(rx)
        return
    subs_o = [O.subchannel(0, 48, eep_level=2, eep_type=0), O.subchannel(100, 58, is_uep=True, uep_index=29)]
    n_frames, cfo, toff, lead = 7, 1.9e-3, -63, 30000
    stream, truth = SM.make_ensemble_stream(O, n_frames, subs_o, seed=41, cfo=cfo, timing_pad=0, noise=2.0, amplitude=1.0)
    NULL, FRAME = O.NB_NULL_PERIOD, O.NB_FRAME_SAMPLES
    rx = C.c_void_p()
    if form == "banked":
        L.dabgpu_receiver_create_banked.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        ck(L.dabgpu_receiver_create_banked(C.byref(rx), 0), "dabgpu_receiver_create_banked")
    else:
        ck(L.dabgpu_receiver_create(C.byref(rx), 0, 1, None, None), "dabgpu_receiver_create")
    try:
        dsubs = (dabgpu.SubChannel * 2)(*[dabgpu.SubChannel(s.start_address, s.length, s.is_uep, s.uep_prot_index, s.eep_prot_level, s.eep_type) for s in subs_o])
        ck(L.dabgpu_receiver_set_subchannels(rx, dsubs, 2, 1), "dabgpu_receiver_set_subchannels")
        ses = L.dabgpu_receiver_session(rx)
        cfg = dabgpu.sync_cfg_default()
        st_o = O.SyncState(0.0, 0.0, 0, 0, 0, 0)
        conj_ref, time_ref = O.sync_refs()
        deint = [O.Deinterleaver(s.length * 8) for s in subs_o]
        pending, expected, undecoded = [], {}, []
        for j in range(n_frames + 1):
            if j < n_frames:
                # the receiver expects the PRS `toff` samples before where it is: NULL | PRS slot | ... of frame j in the staging buffer
                a = lead + j * FRAME - toff
                h, cap = C.c_void_p(), C.c_size_t()
                ck(L.dabgpu_receiver_stage(rx, C.byref(h), C.byref(cap)), "dabgpu_receiver_stage")
                stage = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_float)), shape=(2 * cap.value,)).view(np.complex64)
                seg = stream[a:a + cap.value]
                stage[:seg.size] = seg
                ck(L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), NULL), "dabgpu_receiver_submit_sync")
                rec = dabgpu.SyncState()
                imp = np.zeros(2048, np.float32)
                ck(L.dabgpu_receiver_wait_sync(rx, C.byref(rec), imp.ctypes.data, None), "dabgpu_receiver_wait_sync")
                # oracle: the same frame on the record it carries from frame to frame
                prs_sym = stage[NULL:NULL + 2048].copy()
                O.coarse_freq_sync(prs_sym, st_o, None, time_ref)
                f = np.float32(np.float32(st_o.freq_coarse) + np.float32(st_o.freq_fine))
                ok, off, imp_o = O.fine_time_sync(prs_sym, f, None, conj_ref)
                assert rec.sync_valid == 1 and ok and rec.fine_time_offset == off == toff, (j, rec.fine_time_offset, off)
                assert np.float32(rec.freq_coarse).view(np.uint32) == np.float32(st_o.freq_coarse).view(np.uint32)
                assert np.float32(rec.freq_fine).view(np.uint32) == np.float32(st_o.freq_fine).view(np.uint32)
                assert np.array_equal(imp.view(np.uint32), np.asarray(imp_o, np.float32).view(np.uint32)), "GetImpulseResponse()"
                r = O.demod_frame(stage[NULL + off:NULL + off + FRAME].copy(), f)
                st_o.freq_fine = float(O.update_fine_freq(st_o.freq_fine, r["total_phase"]))
                gen = C.c_uint64()
                if calls == "submit_frame":
                    ck(L.dabgpu_receiver_submit_frame(rx, NULL + off, cfg.fine_freq_update_beta, 0, 0, C.byref(gen)), "dabgpu_receiver_submit_frame")
                else:
                    ck(L.dabgpu_receiver_submit_demod(rx, NULL + off, cfg.fine_freq_update_beta, 0, C.byref(gen)), "dabgpu_receiver_submit_demod")
                    undecoded.append(j)
                    while len(undecoded) > 2:
                        ck(L.dabgpu_receiver_submit_decode(rx, undecoded.pop(0), 0), "dabgpu_receiver_submit_decode")
                assert gen.value == j
                exp = dict(bits=r["bits"], fine=np.float32(st_o.freq_fine), total=np.float32(r["total_phase"]))
                exp["fib"] = [O.fic_decode_group(r["bits"][g * 2304:(g + 1) * 2304], 0) for g in range(4)]
                exp["msc"] = []
                for c in range(4):
                    row = []
                    for si, s in enumerate(subs_o):
                        cif = r["bits"][9216 + c * 55296:9216 + (c + 1) * 55296]
                        deint[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                        lf = deint[si].deinterleave()
                        row.append(None if lf is None else O.msc_decode_logical(s, lf, 0))
                    exp["msc"].append(row)
                expected[j] = exp
                pending.append(j)
            # collect late: three frames in flight
            while pending and (len(pending) > 3 or j == n_frames):
                g = pending.pop(0)
                while undecoded and undecoded[0] <= g:
                    ck(L.dabgpu_receiver_submit_decode(rx, undecoded.pop(0), 0), "dabgpu_receiver_submit_decode")
                fr = Frame()
                ck(L.dabgpu_receiver_wait_frame(rx, g, C.byref(fr)), "dabgpu_receiver_wait_frame")
                e = expected.pop(g)
                bits = np.ctypeslib.as_array(C.cast(fr.bits, C.POINTER(C.c_int8)), shape=(fr.n_bits,))
                assert fr.n_bits == O.NB_FRAME_BITS and np.array_equal(bits, e["bits"]), f"frame {g}: soft bits"
                assert np.float32(fr.freq_fine).view(np.uint32) == e["fine"].view(np.uint32), f"frame {g}: fine frequency word"
                assert np.float32(fr.total_phase).view(np.uint32) == e["total"].view(np.uint32)
                for grp in range(4):
                    out, mask, err = np.zeros(96, np.uint8), C.c_uint32(), C.c_uint64()
                    ck(L.dabgpu_frame_session_fetch_fib_group(ses, g, grp, out.ctypes.data, C.byref(mask), C.byref(err)), "fetch_fib_group")
                    eb, em, ee = e["fib"][grp]
                    assert np.array_equal(out, eb) and mask.value == em and err.value == ee, (g, grp)
                for c in range(4):
                    for si in range(2):
                        if e["msc"][c][si] is None:
                            continue                                     # the time de-interleaver is still filling: the session's output is garbage by contract
                        dec, perr = e["msc"][c][si]
                        out, nb, err = np.zeros(4096, np.uint8), C.c_size_t(), C.c_uint64()
                        ck(L.dabgpu_frame_session_fetch_cif(ses, g, C.byref(dsubs[si]), c, out.ctypes.data, out.size, C.byref(nb), C.byref(err)), "fetch_cif")
                        assert nb.value == dec.size and np.array_equal(out[:nb.value], dec) and err.value == perr, (g, c, si)
        # the last frame's logical frames are what was transmitted 15 CIFs earlier
        # (checked through the oracle's bytes above; here: the oracle's own agree with the generator)
        # ---- a failed impulse-peak test: noise in the PRS slot -> sync_valid 0 -> reset in stream order -> re-acquisition from a zeroed record ----
        h, cap = C.c_void_p(), C.c_size_t()
        ck(L.dabgpu_receiver_stage(rx, C.byref(h), C.byref(cap)), "dabgpu_receiver_stage")
        stage = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_float)), shape=(2 * cap.value,)).view(np.complex64)
        rng = np.random.default_rng(5)
        stage[:] = (rng.standard_normal(cap.value) + 1j * rng.standard_normal(cap.value)).astype(np.complex64)
        ck(L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), NULL), "dabgpu_receiver_submit_sync")
        rec = dabgpu.SyncState()
        ck(L.dabgpu_receiver_wait_sync(rx, C.byref(rec), None, None), "dabgpu_receiver_wait_sync")
        assert rec.sync_valid == 0
        ck(L.dabgpu_receiver_reset(rx), "dabgpu_receiver_reset")
        a = lead + 2 * FRAME - toff
        seg = stream[a:a + cap.value]
        stage[:seg.size] = seg
        ck(L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), NULL), "dabgpu_receiver_submit_sync")
        ck(L.dabgpu_receiver_wait_sync(rx, C.byref(rec), None, None), "dabgpu_receiver_wait_sync")
        fresh = O.SyncState(0.0, 0.0, 0, 0, 0, 0)
        O.coarse_freq_sync(stage[NULL:NULL + 2048].copy(), fresh, None, time_ref)
        assert rec.sync_valid == 1 and rec.fine_time_offset == toff
        assert np.float32(rec.freq_coarse).view(np.uint32) == np.float32(fresh.freq_coarse).view(np.uint32)
        assert np.float32(rec.freq_fine).view(np.uint32) == np.float32(fresh.freq_fine).view(np.uint32)
        # argument checks: a frame outside the staging buffer, a second synchroniser before the first record was collected
        assert L.dabgpu_receiver_submit_frame(rx, cap.value, 0.9, 0, 0, None) == 2
        ck(L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), NULL), "dabgpu_receiver_submit_sync")
        assert L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), NULL) == 2
        ck(L.dabgpu_receiver_wait_sync(rx, C.byref(rec), None, None), "dabgpu_receiver_wait_sync")
    finally:
        L.dabgpu_receiver_destroy(rx)


def test_receivers_come_and_go_without_leaking_device_memory():
    """60 receivers created, used for one synchroniser call and destroyed (each owns two contexts, two streams, a frame session, three
    page-locked staging buffers, events): the device's free memory afterwards is what it was after the first few"""
    import dabgpu
    import torch
    L = _api(dabgpu)
    cfg = dabgpu.sync_cfg_default()
    free = []
    for k in range(60):
        rx = C.c_void_p()
        if k % 3 == 1:                                  # every third one a member of the receiver bank (its slot, result store and events come and go)
            L.dabgpu_receiver_create_banked.argtypes = [C.POINTER(C.c_void_p), C.c_int]
            dabgpu.check(L.dabgpu_receiver_create_banked(C.byref(rx), 0), "dabgpu_receiver_create_banked")
        else:
            dabgpu.check(L.dabgpu_receiver_create(C.byref(rx), 0, 1 + (k % 4 if k % 7 == 0 else 0), None, None), "dabgpu_receiver_create")
        dabgpu.check(L.dabgpu_receiver_submit_sync(rx, C.byref(cfg), 100), "dabgpu_receiver_submit_sync")
        rec = dabgpu.SyncState()
        dabgpu.check(L.dabgpu_receiver_wait_sync(rx, C.byref(rec), None, None), "dabgpu_receiver_wait_sync")
        L.dabgpu_receiver_destroy(rx)
        if k in (9, 59):
            torch.cuda.synchronize()
            free.append(torch.cuda.mem_get_info()[0])
    assert free[1] >= free[0] - (8 << 20), free


def test_decoder_contexts_and_streams_come_and_go_without_leaking():
    """what an MSC_Decoder owns -- a context of its own and a 16-CIF stream on it (dab/msc/msc_decoder.cpp; basic_radio creates one per selected
    service and drops it on deselection) -- created, fed 17 CIFs, decoded and destroyed 120 times: device memory and the process's resident set
    afterwards are what they were after the first twenty"""
    import re
    import dabgpu
    import torch
    L = dabgpu.lib()
    L.dabgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
    L.dabgpu_destroy.argtypes = [C.c_void_p]
    L.dabgpu_msc_stream_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.dabgpu_msc_stream_destroy.argtypes = [C.c_void_p]
    L.dabgpu_msc_stream_push_cif.argtypes = [C.c_void_p, C.c_void_p]
    L.dabgpu_msc_stream_decode_sync.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_int]
    rng = np.random.default_rng(3)
    marks = []
    for k in range(120):
        length = (24, 48, 72, 27)[k % 4]
        sc = dabgpu.SubChannel(0, length, False, 0, 2 if length != 27 else 0, 0 if length != 27 else 1)
        ctx, st = C.c_void_p(), C.c_void_p()
        dabgpu.check(L.dabgpu_create(C.byref(ctx), 0, None, None), "dabgpu_create")
        dabgpu.check(L.dabgpu_msc_stream_create(ctx, C.byref(sc), C.byref(st)), "dabgpu_msc_stream_create")
        cif = rng.integers(-127, 128, length * 64, dtype=np.int8)
        out = np.zeros(length * 8, np.uint8)
        n, err = C.c_size_t(), C.c_uint64()
        for c in range(17):
            dabgpu.check(L.dabgpu_msc_stream_push_cif(st, cif.ctypes.data), "dabgpu_msc_stream_push_cif")
            rc = L.dabgpu_msc_stream_decode_sync(st, out.ctypes.data, C.byref(n), C.byref(err), 0)
            assert rc == (0 if c >= 15 else 4), (k, c, rc)                 # 4 = DABGPU_ERR_NOT_READY: the time de-interleaver is filling
        L.dabgpu_msc_stream_destroy(st)
        L.dabgpu_destroy(ctx)
        if k in (19, 119):
            torch.cuda.synchronize()
            rss = int(re.search(r"VmRSS:\s+(\d+)", open("/proc/self/status").read()).group(1))
            marks.append((torch.cuda.mem_get_info()[0], rss))
    assert marks[1][0] >= marks[0][0] - (8 << 20) and marks[1][1] <= marks[0][1] + 16 * 1024, marks
