"""CPU test of the HOST logic of the drop-in classes (dab-radio_amd/host: OFDM_Demod's reader / delivery threads, the lazy collection of the
synchroniser's record with its replay after a failed impulse-peak test, the frame batcher, FIC_Decoder, MSC_Decoder): tests/cpp/mirror_harness
is linked against tests/cpp/fake_dabgpu_oracle.cpp -- the C ABI implemented by the CPU oracle, no device -- and driven over a hardened
capture (tests/stream_model.py::make_offair_like_capture: echoes, sample-clock error, drop-outs that wipe phase reference symbols) in blocks
of 4,999 / 65,536 / 250,007 samples (a frame spans 40 blocks, 3 blocks, less than one), with and without the frame batcher.  Whatever
the block size, the frames, counters, FIBs and sub-channel bytes must be those of the serial oracle state machine (stream_model.StreamModel,
the restatement of ofdm_demodulator.cpp:235-639) -- the framing may not depend on how far the reader ran ahead of the synchroniser.
(The same classes on the real library: tests/test_gpu_cpp_mirror.py, tests/test_gpu_offair_substitute.py.)"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dab-radio_amd", "host")
CSRC = os.path.join(ROOT, "dab-radio_amd", "csrc")
ORACLE = os.path.join(ROOT, "oracle")
MIRROR_SRCS = ["ofdm/ofdm_demodulator.cpp", "ofdm/dab_refs.cpp", "dab/dabgpu_shared_context.cpp", "dab/dabgpu_frame_batcher.cpp", "dab/fic/fic_decoder.cpp",
               "dab/msc/msc_decoder.cpp", "dab/msc/cif_deinterleaver.cpp", "dab/algorithms/dab_viterbi_decoder.cpp"]
ORACLE_SRCS = ["dab_oracle_ofdm.c", "dab_oracle_decode.c", "dab_oracle_io.c", "dab_oracle_dabplus.c", "dab_oracle_chain.c"]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("mirror_fake")
    objs = []
    for src in ORACLE_SRCS:
        o = d / (src + ".o")
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-mavx2", "-mbmi2", "-mfma", "-c", os.path.join(ORACLE, src), "-o", str(o)],
                       check=True, timeout=600)
        objs.append(str(o))
    exe = d / "mirror_harness_fake"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + ORACLE,
                    os.path.join(ROOT, "tests", "cpp", "mirror_harness.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
                    os.path.join(CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(HOST, s) for s in MIRROR_SRCS] + objs + ["-lm", "-o", str(exe)],
                   check=True, timeout=900)
    return str(exe)


CAPTURES = {
    # 24 frames; the first two drop-outs wipe a NULL + phase reference symbol (failed synchronisation -> NULL search -> re-acquisition)
    "a": dict(n_frames=24, seed=11, dropouts=((5, -0.03, 14000), (12, -0.03, 60000), (17, 0.2, 30000))),
    # two phase reference symbols in a row lost, then most of a frame silent (a NULL search that finds "NULL symbols" inside the gap)
    "b": dict(n_frames=20, seed=12, dropouts=((3, -0.02, 9000), (4, -0.02, 9000), (10, 0.3, 150000))),
}


@pytest.fixture(scope="module")
def captures(tmp_path_factory):
    import oracle as O
    import stream_model as SM
    O.build()
    subs = [O.subchannel(0, 48, eep_level=2, eep_type=0), O.subchannel(120, 27, eep_level=0, eep_type=1)]
    d = tmp_path_factory.mktemp("capture")
    out = dict(O=O, SM=SM, subs=subs)
    for name, c in CAPTURES.items():
        u8, _ = SM.make_offair_like_capture(O, c["n_frames"], subs, seed=c["seed"], dropouts=c["dropouts"])
        iq = O.iq_convert(u8, 0).view(np.complex64)
        iq.tofile(d / f"{name}.c32")
        out[name] = (iq, str(d / f"{name}.c32"))
    return out


@pytest.mark.parametrize("which,block,batch,depth", [("a", 4999, "1", 3), ("a", 65536, "1", 3), ("a", 65536, "0", 1), ("a", 250007, "1", 3), ("a", 700001, "1", 3),
                                                     ("b", 1000, "1", 3), ("b", 30011, "1", 6), ("b", 196608, "1", 3), ("b", 500009, "0", 3)])
def test_frames_do_not_depend_on_the_block_size(harness, captures, tmp_path, which, block, batch, depth):
    """block = 250007 on capture a is the case that found a bug of round 5's pipeline: after a failed impulse-peak test the remainder of the block is
    replayed through the NULL search; when that replay found a NULL symbol and submitted the NEXT synchronisation, its record was collected a
    block later -- with the next block's signal average, and a rewind position that referred to the replayed buffer (OFDM_Demod::CollectPendingSync)."""
    O, SM, subs = captures["O"], captures["SM"], captures["subs"]
    iq, path = captures[which]
    model = SM.StreamModel(O)
    for k in range(0, iq.size, block):
        model.process(iq[k:k + block])
    frames = [f["bits"] for f in model.out_frames]
    nf = len(frames)
    assert nf >= CAPTURES[which]["n_frames"] - 7 and model.frames_desync >= 2, "the capture must exercise re-acquisition"
    fibs, msc = SM.expected_decode(O, frames, subs)

    out = tmp_path / "out"
    out.mkdir()
    args = [harness, path, str(out), str(block), "0", "48", "2", "0", "120", "27", "0", "1"]
    res = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, DABGPU_MIRROR_BATCH=batch, DABGPU_MIRROR_DEPTH=str(depth)), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert f"frames={nf} read={nf} desync={model.frames_desync} state={model.state}" in res.stdout, res.stdout
    bits = np.fromfile(out / "frame_bits.bin", dtype=np.int8).reshape(nf, O.NB_FRAME_BITS)
    states = np.fromfile(out / "states.bin", dtype=np.float32).reshape(nf, 4)
    for k, fr in enumerate(model.out_frames):
        assert int(states[k, 2]) == fr["offset"] and int(states[k, 3]) == fr["desync"], f"frame {k}: fine time offset / desync count"
        assert states[k, 0].view(np.uint32) == np.float32(fr["coarse"]).view(np.uint32) and states[k, 1].view(np.uint32) == np.float32(fr["fine"]).view(np.uint32)
        assert np.array_equal(bits[k], frames[k]), f"frame {k} soft bits"
    assert (out / "fibs.bin").read_bytes() == bytes(fibs)
    # msc_<k>.bin: per CIF a uint32 length + the bytes (length 0 while the time de-interleaver fills)
    for si in range(len(subs)):
        raw = (out / f"msc_{si}.bin").read_bytes()
        got, pos = bytearray(), 0
        while pos < len(raw):
            n = int(np.frombuffer(raw[pos:pos + 4], np.uint32)[0])
            got += raw[pos + 4:pos + 4 + n]
            pos += 4 + n
        assert bytes(got) == bytes(msc[si]), f"sub-channel {si}"


REAL_HARNESS = os.path.join(ROOT, "tests", "cpp", "mirror_harness")


@pytest.mark.gpu
@pytest.mark.parametrize("which,seed,depth", [("a", 11, 3), ("b", 12, 1), ("a", 13, 6), ("b", 14, 3)])
def test_random_block_lengths_and_caller_resets_on_the_device(captures, tmp_path, which, seed, depth):
    """the same schedules through the real library: the synchroniser's record really is in flight when the next block (or the caller's Reset()) arrives"""
    if not os.path.exists(REAL_HARNESS):
        import __graft_entry__ as g
        g.build()
    env = {"LD_LIBRARY_PATH": os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""), "DABGPU_MIRROR_DEPTH": str(depth)}
    check_random_schedule(REAL_HARNESS, captures, tmp_path, which, seed, env)


@pytest.mark.parametrize("which,seed", [("a", 1), ("a", 2), ("b", 3), ("b", 4), ("a", 5), ("b", 6)])
def test_random_block_lengths_and_caller_resets(harness, captures, tmp_path, which, seed):
    check_random_schedule(harness, captures, tmp_path, which, seed, {})


def check_random_schedule(harness, captures, tmp_path, which, seed, env_extra):
    """every Process() call gets its own length (log-uniform 100 .. 600,000 samples), and now and then the caller calls Reset() between two calls
    (ofdm_demodulator.cpp:277-289: the GUI's button) -- also while a synchroniser's record is still in flight, which the serial machine had
    long acted on: frames, counters and decoded bytes stay those of the serial machine fed the same calls"""
    O, SM, subs = captures["O"], captures["SM"], captures["subs"]
    iq, path = captures[which]
    rng = np.random.default_rng(1000 + seed)
    schedule, total = [], 0
    while total < iq.size:
        n = int(np.exp(rng.uniform(np.log(100.0), np.log(600000.0))))
        if rng.random() < 0.04 and total > 0:
            n = -n
        schedule.append(n)
        total += abs(n)
    (tmp_path / "schedule.txt").write_text("\n".join(str(v) for v in schedule) + "\n")
    model = SM.StreamModel(O)
    pos = 0
    for v in schedule:
        if pos >= iq.size:
            break
        if v < 0:
            model.reset()
        model.process(iq[pos:pos + abs(v)])
        pos += abs(v)
    frames = [f["bits"] for f in model.out_frames]
    nf = len(frames)
    assert nf >= 8
    fibs, msc = SM.expected_decode(O, frames, subs)
    out = tmp_path / "out"
    out.mkdir()
    args = [harness, path, str(out), "65536", "0", "48", "2", "0", "120", "27", "0", "1"]
    res = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, DABGPU_HARNESS_SCHEDULE=str(tmp_path / "schedule.txt"), **env_extra), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert f"frames={nf} read={nf} desync={model.frames_desync} state=" in res.stdout, (res.stdout, nf, model.frames_desync)
    bits = np.fromfile(out / "frame_bits.bin", dtype=np.int8).reshape(nf, O.NB_FRAME_BITS)
    states = np.fromfile(out / "states.bin", dtype=np.float32).reshape(nf, 4)
    for k, fr in enumerate(model.out_frames):
        assert int(states[k, 2]) == fr["offset"] and int(states[k, 3]) == fr["desync"], f"frame {k}: fine time offset / desync count"
        assert np.array_equal(bits[k], frames[k]), f"frame {k} soft bits"
    assert (out / "fibs.bin").read_bytes() == bytes(fibs)


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_other_transmission_modes_with_random_block_lengths(harness, tmp_path, mode):
    """the reader's framing in the geometries of modes II-IV (dab_ofdm_params_ref.cpp:22-60) -- earliest frame end, staging capacity, NULL search window --
    on a noisy stream where the short phase reference symbols of modes II / III fail the impulse-peak test now and then: random Process() lengths
    and caller resets, against the serial machine"""
    import oracle as O
    import modes_model as MM
    import stream_model as SM
    O.build()
    rng = np.random.default_rng(40 + mode)
    g = O.geometry(mode)
    sent = [rng.integers(0, 2, g.nb_frame_bits, dtype=np.uint8) for _ in range(10)]
    tx = O.apply_pll(np.concatenate([MM.make_tx_frame(O, mode, b, rng) for b in sent]), 2.1 / g.nb_fft, 0.2)
    stream = np.concatenate([tx[g.nb_null_period:g.nb_null_period + 6000 + 77], tx])
    stream = ((stream + 0.05 * (rng.standard_normal(stream.size) + 1j * rng.standard_normal(stream.size))) / 39.2).astype(np.complex64)
    stream.tofile(tmp_path / "iq.c32")
    schedule, total = [], 0
    while total < stream.size:
        n = int(np.exp(rng.uniform(np.log(100.0), np.log(3.0 * g.nb_frame_samples))))
        if rng.random() < 0.04 and total > 0:
            n = -n
        schedule.append(n)
        total += abs(n)
    (tmp_path / "schedule.txt").write_text("\n".join(str(v) for v in schedule) + "\n")
    model = SM.StreamModel(O, mode)
    model.cfg.impulse_peak_threshold_db = 8.0
    pos = 0
    for v in schedule:
        if pos >= stream.size:
            break
        if v < 0:
            model.reset()
        model.process(stream[pos:pos + abs(v)])
        pos += abs(v)
    nf = len(model.out_frames)
    assert nf >= 4
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, DABGPU_HARNESS_MODE=str(mode), DABGPU_HARNESS_PEAK_DB="8", DABGPU_HARNESS_SCHEDULE=str(tmp_path / "schedule.txt"))
    res = subprocess.run([harness, str(tmp_path / "iq.c32"), str(out), "16384"], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert f"frames={nf} read={nf} desync={model.frames_desync} state={model.state}" in res.stdout, (res.stdout, nf, model.frames_desync, model.state)
    bits = np.fromfile(out / "frame_bits.bin", dtype=np.int8).reshape(nf, g.nb_frame_bits)
    states = np.fromfile(out / "states.bin", dtype=np.float32).reshape(nf, 4)
    for k, fr in enumerate(model.out_frames):
        assert int(states[k, 2]) == fr["offset"] and int(states[k, 3]) == fr["desync"], (mode, k)
        assert np.array_equal(bits[k], fr["bits"]), (mode, k)
