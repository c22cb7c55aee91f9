"""End-to-end -m gpu test of the C++ mirror classes (dab-radio_amd/host: OFDM_Demod, FIC_Decoder, MSC_Decoder,
CIF_Deinterleaver, DAB_Viterbi_Decoder) driven by tests/cpp/mirror_harness exactly like basic_radio_app drives the
reference's classes, against the CPU oracle composed in tests/stream_model.py on the same impaired IQ stream
(BASELINE config 1 substitute: CFO + timing offset + AWGN, no off-air recording is available offline).
Bar: every produced byte identical -- frame soft bits, FIBs, FIB CRC masks, path errors, sub-channel bytes,
de-interleaved logical frames, raw Viterbi output, frequency offsets (float32 bit patterns) and time offsets."""
import os
import subprocess

import numpy as np
import pytest

import stream_model as SM_CORE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "mirror_harness")


@pytest.mark.parametrize("batch", ["1", "0"], ids=["frame_batcher", "call_by_call"])
@pytest.mark.parametrize("decode_threads", [1, 2], ids=["one_thread", "thread_per_sub_channel"])
def test_cpp_mirror_stream_matches_oracle(oracle, tmp_path, decode_threads, batch):
    """decode_threads = 2: the two MSC_Decoders of the harness are driven from two threads at once, as basic_radio's thread pool
    drives one task per sub-channel (each decoder owns a device context): same files, byte for byte.
    batch: DABGPU_MIRROR_BATCH -- with the frame batcher (default) OFDM_Demod hands every frame to one batched device decode and the
    classes pick their bytes up once 16 consecutive CIFs went through it (from CIF 16 of this stream on); without it every
    DecodeFIBGroup / DecodeCIF is its own launch.  Identical files either way."""
    import stream_model as SM
    if not os.path.exists(HARNESS):
        import __graft_entry__ as g
        g.build()
    subs = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(120, 27, eep_level=0, eep_type=1)]
    n_frames = 7
    stream, truth = SM.make_ensemble_stream(oracle, n_frames, subs, seed=2025)
    iq_path = tmp_path / "iq.c32"
    stream.tofile(iq_path)
    out = tmp_path / "out"
    out.mkdir()
    args = [HARNESS, str(iq_path), str(out), "65536"]
    for s in subs:
        args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
    env = dict(os.environ, DABGPU_HARNESS_THREADS=str(decode_threads), DABGPU_MIRROR_BATCH=batch)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]

    # ---- expected side: oracle state machine on the same blocks ----
    model = SM.StreamModel(oracle)
    for k in range(0, stream.size, 65536):
        model.process(stream[k:k + 65536])
    nf = len(model.out_frames)
    assert nf >= n_frames - 1, "the stream must yield (almost) every transmitted frame"
    assert f"frames={nf} read={nf} desync={model.frames_desync} state={model.state}" in res.stdout, res.stdout

    bits = np.fromfile(out / "frame_bits.bin", dtype=np.int8).reshape(nf, oracle.NB_FRAME_BITS)
    states = np.fromfile(out / "states.bin", dtype=np.float32).reshape(nf, 4)
    fft1 = np.fromfile(out / "fft_sym1.bin", dtype=np.complex64).reshape(nf, 2048)
    dq0 = np.fromfile(out / "dqpsk_sym0.bin", dtype=np.complex64).reshape(nf, 1536)
    for k, fr in enumerate(model.out_frames):
        assert np.array_equal(bits[k], fr["bits"]), f"frame {k} soft bits"
        assert states[k, 0].view(np.uint32) == np.float32(fr["coarse"]).view(np.uint32)
        assert states[k, 1].view(np.uint32) == np.float32(fr["fine"]).view(np.uint32)
        assert int(states[k, 2]) == fr["offset"] and int(states[k, 3]) == fr["desync"]
        assert np.array_equal(fft1[k].view(np.uint32), fr["fft"][2048:4096].view(np.uint32)), f"frame {k} GetFrameFFT"
        X0, X1 = fr["fft"][:2048], fr["fft"][2048:4096]
        bins = np.r_[2048 - 768:2048, 1:769]
        exp_dq = np.array([complex(*oracle_conj_mul(oracle, X0[b], X1[b])) for b in bins[:8]], dtype=np.complex64)
        assert np.array_equal(dq0[k][:8].view(np.uint32), exp_dq.view(np.uint32)), f"frame {k} GetFrameDataVec"

    # ---- FIC ----
    st = np.fromfile(out / "fic_status.bin", dtype=np.uint8).reshape(nf * 4, 12)
    fib_stream = np.fromfile(out / "fibs.bin", dtype=np.uint8) if (out / "fibs.bin").exists() else np.zeros(0, np.uint8)
    vraw = np.fromfile(out / "viterbi_raw.bin", dtype=np.uint8).reshape(nf, 96)
    vmeta = np.fromfile(out / "viterbi_meta.bin", dtype=np.uint64).reshape(nf, 3)
    exp_fibs = []
    n_valid = 0
    for k, fr in enumerate(model.out_frames):
        for g in range(4):
            eb, em, ee = oracle.fic_decode_group(fr["bits"][g * 2304:(g + 1) * 2304], SM_CORE.mirror_core_model())
            assert int(st[k * 4 + g, :4].view(np.uint32)[0]) == em and int(st[k * 4 + g, 4:].view(np.uint64)[0]) == ee
            for i in range(3):
                if em & (1 << i):
                    exp_fibs.append(eb[32 * i:32 * i + 30])
                    n_valid += 1
            if g == 0:
                assert np.array_equal(vraw[k] ^ oracle.scrambler_bytes(96), eb), "DAB_Viterbi_Decoder raw output"
                assert list(vmeta[k]) == [2304, 774, ee]
    assert n_valid >= 12 * (nf - 1), "at this SNR every FIB after the acquisition frame passes its CRC"
    assert np.array_equal(fib_stream, np.concatenate(exp_fibs))

    # ---- MSC + CIF_Deinterleaver ----
    deint_ref = [oracle.Deinterleaver(s.length * 8) for s in subs]
    exp_msc = [bytearray() for _ in subs]
    exp_deint = bytearray()
    n_msc_ok = 0
    first_tx_frame = 0                        # output frame k is transmitted frame k (only the unfinished last one is lost)
    for k, fr in enumerate(model.out_frames):
        for c in range(4):
            cif = fr["bits"][9216 + c * 55296: 9216 + (c + 1) * 55296]
            for si, s in enumerate(subs):
                deint_ref[si].consume(cif[s.start_address * 64:(s.start_address + s.length) * 64])
                lf = deint_ref[si].deinterleave()
                if lf is None:
                    exp_msc[si] += np.uint32(0).tobytes()
                else:
                    dec, _ = oracle.msc_decode_logical(s, lf, SM_CORE.mirror_core_model())
                    exp_msc[si] += np.uint32(dec.size).tobytes() + dec.tobytes()
                    t = 4 * (k + first_tx_frame) + c - 15
                    n_msc_ok += int(np.array_equal(dec, truth["payload"][si][t]))
                if si == 0:
                    exp_deint += (b"\x00" if lf is None else b"\x01" + lf.tobytes())
    for si in range(len(subs)):
        assert (out / f"msc_{si}.bin").read_bytes() == bytes(exp_msc[si]), f"sub-channel {si}"
    assert (out / "deint.bin").read_bytes() == bytes(exp_deint)
    # the first frame is demodulated with the acquisition's coarse timing (often garbage); logical frames that only draw
    # on later CIFs must decode to the transmitted payload
    assert n_msc_ok >= len(subs) * (4 * nf - 19), "decoded sub-channel bytes must be the transmitted payload"


def oracle_conj_mul(oracle, a, b):
    import ctypes as C

    class CF(C.Structure):
        _fields_ = [("re", C.c_float), ("im", C.c_float)]
    L = oracle.lib()
    L.dab_conj_mul.restype = CF
    L.dab_conj_mul.argtypes = [CF, CF]
    r = L.dab_conj_mul(CF(float(a.real), float(a.imag)), CF(float(b.real), float(b.imag)))
    return r.re, r.im


def test_two_receivers_with_reader_radio_and_worker_threads_on_the_device(oracle, tmp_path):
    """tests/cpp/mirror_threads_driver on the real library: two OFDM_Demod receivers (each with its receiver pipeline, delivery thread and
    frame session) in ONE process, first one after the other from one thread, then both at once with a reader thread, a radio thread and
    two decode workers each -- the way basic_radio_app runs its OFDM and radio threads (examples/basic_radio_app.cpp:404-419).  The threaded
    bytes (FIBs, sub-channel bytes, frame counts) must equal the serial ones receiver by receiver, and the two receivers' outputs differ.
    (The same driver runs under ThreadSanitizer / ASan against the oracle-backed fake ABI in tests/test_host_sanitizers.py.)"""
    import json
    import stream_model as SM
    driver = os.path.join(ROOT, "tests", "cpp", "mirror_threads_driver")
    if not os.path.exists(driver):
        import __graft_entry__ as g
        g.build()
    subs = [oracle.subchannel(0, 24, eep_level=2, eep_type=0), oracle.subchannel(60, 21, eep_level=1, eep_type=1)]
    paths = []
    for k in range(2):
        stream, _ = SM.make_ensemble_stream(oracle, 9, subs, seed=900 + k, cfo=(1.1e-3, -2.4e-3)[k], timing_pad=(300, 4321)[k], noise=2.0)
        p = tmp_path / f"rx{k}.c32"
        stream.tofile(p)
        paths.append(str(p))
    args = [driver, "65536"]
    for s in subs:
        args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.pop("DABGPU_MIRROR_BANK", None)
    digests = {}
    # private pipelines only / every receiver a member of the receiver bank (csrc/receiver_bank.hip: one synchroniser launch, one demodulation launch
    # and one decode for what both posted) / a mix (the second receiver joins the bank, the first keeps its pipeline): the same bytes
    env.pop("DABGPU_MIRROR_BANK_FROM", None)
    for bank in ("0", "1", None):
        # (None: the AUTO rule with its threshold lowered to the second receiver -- one private pipeline and one bank member side by side)
        e = dict(env, DABGPU_MIRROR_BANK_FROM="2") if bank is None else dict(env, DABGPU_MIRROR_BANK=bank)
        res = subprocess.run(args + ["--"] + paths, capture_output=True, text=True, env=e, timeout=600)
        assert res.returncode == 0, (bank, res.stdout[-2000:], res.stderr[-2000:])
        out = json.loads(res.stdout.strip().splitlines()[-1])
        assert out["ok"] and out["receivers"] == 2, bank
        for r in out["per_receiver"]:
            assert r["frames"] >= 7 and r["fib_bytes"] >= 30 * 12 * (r["frames"] - 2) and r["cifs_with_output"] >= 2 * (4 * r["frames"] - 15) and r["threaded_equals_serial"], bank
        assert out["per_receiver"][0]["digest"] != out["per_receiver"][1]["digest"]
        digests[bank] = [(r["digest"], r["frames"], r["fib_bytes"], r["cifs_with_output"]) for r in out["per_receiver"]]
    assert digests["1"] == digests["0"] and digests[None] == digests["0"], digests


def test_eight_banked_receivers_equal_eight_private_ones(oracle, tmp_path):
    """eight receivers of one process, reader + radio + worker threads each, every one on its own capture (carrier offset, timing, payload): all of
    them members of the receiver bank against all of them on private pipelines -- digest of FIBs + sub-channel bytes, frames and counters equal,
    receiver by receiver; and the serial phase of the driver (one receiver at a time: rounds of one job) equals the threaded one (full rounds)"""
    import json
    import stream_model as SM
    driver = os.path.join(ROOT, "tests", "cpp", "mirror_threads_driver")
    if not os.path.exists(driver):
        import __graft_entry__ as g
        g.build()
    subs = [oracle.subchannel(0, 48, eep_level=2, eep_type=0), oracle.subchannel(100, 58, is_uep=True, uep_index=29), oracle.subchannel(300, 42, eep_level=1, eep_type=1)]
    paths = []
    for k in range(8):
        stream, _ = SM.make_ensemble_stream(oracle, 8, subs[:2] + [subs[2]], seed=1200 + k, cfo=(-2.2e-3 + 0.6e-3 * k), timing_pad=137 * k + 11, noise=2.0)
        p = tmp_path / f"rx{k}.c32"
        stream.tofile(p)
        paths.append(str(p))
    args = [driver, "65536"]
    for s in subs:
        if s.is_uep:
            continue                                   # (the driver's command line takes EEP sub-channels; the UEP one above still shapes the multiplex)
        args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for k in ("DABGPU_MIRROR_BANK", "DABGPU_MIRROR_BANK_FROM", "DABGPU_BANK_ROUNDS", "DABGPU_BANK_GATHER_US"):
        env.pop(k, None)
    digests = {}
    # private pipelines / the bank as it comes (at most two rounds under way) / one round at a time, no gathering window (every round carries whatever
    # is queued: up to eight frames) / eight rounds deep (rounds of one or two frames) / the classes' own rule (one private pipeline, seven members)
    cases = {"0": {"DABGPU_MIRROR_BANK": "0"}, "1": {"DABGPU_MIRROR_BANK": "1"},
             "one round": {"DABGPU_MIRROR_BANK": "1", "DABGPU_BANK_ROUNDS": "1", "DABGPU_BANK_GATHER_US": "0"},
             "eight rounds": {"DABGPU_MIRROR_BANK": "1", "DABGPU_BANK_ROUNDS": "8"}, "auto": {}}
    for bank, extra in cases.items():
        res = subprocess.run(args + ["--"] + paths, capture_output=True, text=True, env=dict(env, **extra), timeout=900)
        assert res.returncode == 0, (bank, res.stdout[-2000:], res.stderr[-2000:])
        out = json.loads(res.stdout.strip().splitlines()[-1])
        assert out["ok"] and out["receivers"] == 8, bank
        assert all(r["threaded_equals_serial"] and r["frames"] >= 6 for r in out["per_receiver"]), bank
        digests[bank] = [(r["digest"], r["frames"], r["fib_bytes"], r["cifs_with_output"]) for r in out["per_receiver"]]
    for bank in cases:
        assert digests[bank] == digests["0"], bank
    assert len({d[0] for d in digests["0"]}) == 8
