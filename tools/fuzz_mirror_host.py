#!/usr/bin/env python3
"""Fuzz of the HOST logic of the drop-in classes, no GPU: tests/cpp/mirror_harness linked with the oracle-backed C ABI (tests/cpp/fake_dabgpu_oracle.cpp)
is fed three hardened captures (drop-outs that wipe phase reference symbols) with a random length for every Process() call (three ranges) and
caller-side Reset() calls in 5 % of the gaps, at 1-4 frames in flight, and compared with the serial oracle state machine
(tests/stream_model.py::StreamModel) fed the same calls: frames, soft bits, fine time offsets, desync counts, final state.
tests/test_mirror_host_logic.py runs six such schedules; this runs as many as asked.

    python tools/fuzz_mirror_host.py FIRST_SEED LAST_SEED        # 260 seeds were clean in round 5 after the CollectPendingSync fix
"""
import os, sys, subprocess, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import oracle as O, stream_model as SM
import test_mirror_host_logic as T
O.build()
W = tempfile.mkdtemp(prefix="fuzz_mirror_")
objs = []
for src in T.ORACLE_SRCS:
    o = os.path.join(W, src + ".o")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-mavx2", "-mbmi2", "-mfma", "-c", os.path.join(T.ORACLE, src), "-o", o], check=True)
    objs.append(o)
EXE = os.path.join(W, "mirror_harness_fake")
subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + T.HOST, "-I" + os.path.join(ROOT, "include"), "-I" + T.CSRC, "-I" + T.ORACLE,
                os.path.join(ROOT, "tests", "cpp", "mirror_harness.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
                os.path.join(T.CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(T.HOST, s) for s in T.MIRROR_SRCS] + objs + ["-lm", "-o", EXE], check=True)
subs = [O.subchannel(0, 48, eep_level=2, eep_type=0)]
caps = {}
for name, c in {"a": dict(n_frames=24, seed=11, dropouts=((5, -0.03, 14000), (12, -0.03, 60000), (17, 0.2, 30000))),
                "b": dict(n_frames=20, seed=12, dropouts=((3, -0.02, 9000), (4, -0.02, 9000), (10, 0.3, 150000))),
                "c": dict(n_frames=16, seed=13, dropouts=((2, -0.03, 20000), (5, -0.03, 20000), (8, -0.03, 20000), (11, -0.03, 20000)))}.items():
    u8, _ = SM.make_offair_like_capture(O, c["n_frames"], subs, seed=c["seed"], dropouts=c["dropouts"])
    iq = O.iq_convert(u8, 0).view(np.complex64)
    iq.tofile(os.path.join(W, name + '.c32')); caps[name] = iq
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    which = "abc"[seed % 3]
    iq = caps[which]
    rng = np.random.default_rng(seed)
    lo, hi = [(100, 600000), (2000, 30000), (50000, 400000)][(seed // 3) % 3]
    schedule, total = [], 0
    while total < iq.size:
        n = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        if rng.random() < 0.05 and total > 0: n = -n
        schedule.append(n); total += abs(n)
    open(os.path.join(W, 'schedule.txt'),'w').write("\n".join(map(str, schedule)) + "\n")
    model = SM.StreamModel(O); pos = 0
    for v in schedule:
        if pos >= iq.size: break
        if v < 0: model.reset()
        model.process(iq[pos:pos + abs(v)]); pos += abs(v)
    nf = len(model.out_frames)
    shutil.rmtree(os.path.join(W, 'out'), ignore_errors=True); os.makedirs(os.path.join(W, 'out'))
    res = subprocess.run([EXE, os.path.join(W, which + '.c32'), os.path.join(W, 'out'), '65536', "0", "48", "2", "0"], capture_output=True, text=True,
                         env=dict(os.environ, DABGPU_HARNESS_SCHEDULE=os.path.join(W, 'schedule.txt'), DABGPU_MIRROR_DEPTH=str(1 + seed % 4)))
    ok = res.returncode == 0 and f"frames={nf} read={nf} desync={model.frames_desync} state={model.state}" in res.stdout
    if ok and nf:
        bits = np.fromfile(os.path.join(W, 'out', 'frame_bits.bin'), dtype=np.int8).reshape(-1, O.NB_FRAME_BITS)
        states = np.fromfile(os.path.join(W, 'out', 'states.bin'), dtype=np.float32).reshape(-1, 4)
        ok = bits.shape[0] == nf and all(np.array_equal(bits[k], fr["bits"]) and int(states[k,2]) == fr["offset"] and int(states[k,3]) == fr["desync"] for k, fr in enumerate(model.out_frames))
    print(seed, which, nf, model.frames_desync, model.state, "OK" if ok else "MISMATCH " + res.stdout.strip()[-120:] + res.stderr[-200:], flush=True)
    bad += not ok
    if not ok: shutil.copy(os.path.join(W, 'schedule.txt'), f'schedule_bad_{seed}.txt')
print("bad", bad)
shutil.rmtree(W, ignore_errors=True)
sys.exit(1 if bad else 0)
