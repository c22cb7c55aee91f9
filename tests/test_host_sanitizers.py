"""CPU sanitizers and fuzzing for everything that is not a kernel (sanitizers belong on the CPU build; the reference ships an
AddressSanitizer preset, CMakePresets.json:47-53):

 (a) the CPU ORACLE -- the checker that grades everything -- built with -fsanitize=address,undefined (oracle/Makefile SAN=1) runs its
     own CPU tests clean;
 (b) the DEVICE-FREE PART OF libdabgpu.so (dab-radio_amd/csrc/dabgpu_host_logic.cpp: protection-profile plans, codeword validation,
     mapping cost model, run-length rules, constant tables, capture-format and wav-header parsing), built on its own under ASan + UBSan
     and fuzzed (tests/cpp/host_logic_fuzz.cpp): hostile sub-channel descriptors, wav images with lying chunk sizes, truncations;
 (c) the C++ MIRROR CLASSES' host code (framing state machine, frame batcher with one session per demodulator, shared context,
     decoders) under ThreadSanitizer and under ASan + UBSan, two receivers in one process with reader / radio / worker threads
     (tests/cpp/mirror_threads_driver.cpp), linked against a TEST-ONLY implementation of the C ABI entry points on the oracle
     (tests/cpp/fake_dabgpu_oracle.cpp -- under tests/, never part of the product)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dab-radio_amd", "csrc")
HOST = os.path.join(ROOT, "dab-radio_amd", "host")
ORACLE = os.path.join(ROOT, "oracle")


def lib_of(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip(f"{name} is not installed with this gcc")
    return p


def run(cmd, **kw):
    res = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert res.returncode == 0, (" ".join(map(str, cmd)), res.stdout[-3000:], res.stderr[-6000:])
    return res


def test_oracle_is_clean_under_address_and_undefined_behaviour_sanitizers():
    asan, ubsan = lib_of("libasan.so"), lib_of("libubsan.so")
    run(["make", "-C", ORACLE, "SAN=1", "libdab_oracle_san.so"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", DAB_ORACLE_SO=os.path.join(ORACLE, "libdab_oracle_san.so"))
    tests = ["test_oracle_pins.py", "test_oracle_properties.py", "test_oracle_chain.py", "test_oracle_dabplus.py", "test_oracle_modes.py",
             "test_io_formats.py", "test_independent_pins.py"]
    res = run(["python", "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", t) for t in tests], env=env, timeout=1500)
    assert " passed" in res.stdout and "failed" not in res.stdout, res.stdout[-2000:]


def test_device_free_library_code_fuzzed_under_asan_and_ubsan(tmp_path):
    lib_of("libasan.so")
    exe = tmp_path / "host_logic_fuzz"
    run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(ROOT, "tests", "cpp", "host_logic_fuzz.cpp"),
         os.path.join(CSRC, "dabgpu_host_logic.cpp"), "-o", str(exe)], timeout=600)
    for seed in (1, 2, 3):
        res = run([str(exe), "30000", str(seed)], env=dict(os.environ, ASAN_OPTIONS="abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"), timeout=600)
        out = json.loads(res.stdout.strip().splitlines()[-1])
        assert out["failed_checks"] == 0
        # the fuzzer reached both sides of every decision: it saw accepted AND rejected inputs of each kind
        assert 0 < out["accepted_plans"] < out["iterations"] and 0 < out["accepted_wav"] < out["iterations"] and 0 < out["accepted_codewords"] < out["iterations"]


MIRROR_SRCS = ["ofdm/ofdm_demodulator.cpp", "ofdm/dab_refs.cpp", "dab/dabgpu_shared_context.cpp", "dab/dabgpu_frame_batcher.cpp", "dab/fic/fic_decoder.cpp",
               "dab/msc/msc_decoder.cpp"]
ORACLE_SRCS = ["dab_oracle_ofdm.c", "dab_oracle_decode.c", "dab_oracle_io.c", "dab_oracle_dabplus.c", "dab_oracle_chain.c"]


def build_driver(tmp_path, tag, san_flags):
    """mirror classes + device-free library code + the fake ABI + the oracle, everything instrumented, into one executable"""
    objs = []
    for src in ORACLE_SRCS:
        o = tmp_path / f"{tag}_{src}.o"
        run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-DDAB_ORACLE_NO_CLONES"] + san_flags + ["-c", os.path.join(ORACLE, src), "-o", str(o)], timeout=600)
        objs.append(str(o))
    exe = tmp_path / f"mirror_threads_{tag}"
    run(["g++", "-O1", "-g", "-std=c++17", "-pthread"] + san_flags +
        ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + ORACLE,
         os.path.join(ROOT, "tests", "cpp", "mirror_threads_driver.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
         os.path.join(CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(HOST, s) for s in MIRROR_SRCS] + objs + ["-lm", "-o", str(exe)], timeout=900)
    return exe


@pytest.fixture(scope="module")
def two_receiver_streams(tmp_path_factory):
    """two different ensembles (own payload, carrier offset, timing) as complex-float capture files; the multiplex layout is shared"""
    import oracle as O
    import stream_model as SM
    O.build()
    d = tmp_path_factory.mktemp("iq")
    subs = [O.subchannel(0, 24, eep_level=2, eep_type=0), O.subchannel(60, 21, eep_level=1, eep_type=1)]
    paths = []
    for k in range(2):
        stream, _ = SM.make_ensemble_stream(O, 7, subs, seed=900 + k, cfo=(1.1e-3, -2.4e-3)[k], timing_pad=(300, 4321)[k], noise=2.0)
        p = d / f"rx{k}.c32"
        stream.tofile(p)
        paths.append(str(p))
    return subs, paths


@pytest.mark.parametrize("tag,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
     {"ASAN_OPTIONS": "abort_on_error=1:detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}),
])
def test_mirror_classes_with_two_receivers_and_decoder_threads(tmp_path, two_receiver_streams, tag, flags, env):
    lib_of("libtsan.so" if tag == "tsan" else "libasan.so")
    subs, paths = two_receiver_streams
    exe = build_driver(tmp_path, tag, flags)
    args = [str(exe), "65536"]
    for s in subs:
        args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
    res = run(args + ["--"] + paths, env=dict(os.environ, **env), timeout=1200)
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["receivers"] == 2 and "ThreadSanitizer" not in res.stderr and "AddressSanitizer" not in res.stderr
    for r in out["per_receiver"]:
        # every transmitted frame but the first was demodulated, its FIBs passed their CRCs, and the time de-interleaver produced
        # logical frames from the 16th CIF on -- through the batcher's sessions where the decoders had caught up with them
        assert r["frames"] >= 5 and r["fib_bytes"] >= 30 * 12 * (r["frames"] - 2) and r["cifs_with_output"] >= 2 * (4 * r["frames"] - 15) and r["threaded_equals_serial"]
    assert out["per_receiver"][0]["digest"] != out["per_receiver"][1]["digest"]


def test_decoders_created_and_dropped_on_the_delivery_thread_under_thread_sanitizer(tmp_path):
    """tests/cpp/mirror_lifecycle_driver (decoders created / destroyed inside the frame observer, i.e. on OFDM_Demod's delivery thread, while the reader
    thread asks the frame batcher what is listened to and submits frames) with the "churn" script of tests/test_mirror_lifecycle.py, instrumented"""
    lib_of("libtsan.so")
    import oracle as O
    import stream_model as SM
    import test_mirror_lifecycle as L
    O.build()
    subs = [O.subchannel(v[0], v[1], eep_level=v[2], eep_type=v[3]) for v in L.SUBS.values()]
    iq, _ = SM.make_ensemble_stream(O, 18, subs, seed=78)
    iq.tofile(tmp_path / "iq.c32")
    lines = []
    for fr, op, ident in L.SCRIPTS["churn"]:
        lines.append(f"{fr} add {ident} {L.SUBS[ident][0]} {L.SUBS[ident][1]} {L.SUBS[ident][2]} {L.SUBS[ident][3]}" if op == "add" else f"{fr} {op} {ident}")
    (tmp_path / "script.txt").write_text("\n".join(lines) + "\n")
    san = ["-fsanitize=thread"]
    objs = []
    for src in ORACLE_SRCS:
        o = tmp_path / f"lc_{src}.o"
        run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-DDAB_ORACLE_NO_CLONES"] + san + ["-c", os.path.join(ORACLE, src), "-o", str(o)], timeout=600)
        objs.append(str(o))
    exe = tmp_path / "mirror_lifecycle_tsan"
    run(["g++", "-O1", "-g", "-std=c++17", "-pthread"] + san + ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + ORACLE,
         os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
         os.path.join(CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(HOST, s) for s in MIRROR_SRCS + ["dab/msc/cif_deinterleaver.cpp", "dab/algorithms/dab_viterbi_decoder.cpp"]] +
        objs + ["-lm", "-o", str(exe)], timeout=900)
    out = tmp_path / "out"
    out.mkdir()
    res = run([str(exe), str(tmp_path / "iq.c32"), str(out), "65536", str(tmp_path / "script.txt")],
              env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"), timeout=1200)
    assert "ThreadSanitizer" not in res.stderr, res.stderr[-3000:]
    assert "frames=1" in res.stdout and "cifs_batched=" in res.stdout, res.stdout
