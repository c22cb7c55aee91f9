#!/usr/bin/env python3
"""Fuzz of the drop-in classes' host logic with decoders that come and go, no GPU: tests/cpp/mirror_lifecycle_driver linked with the oracle-backed ABI on the
hardened captures of tests/test_mirror_host_logic.py (drop-outs: losses of lock and re-acquisitions between the frames), a random life-cycle script
(tests/test_mirror_lifecycle.py::random_script), a random length for every Process() call, caller resets, 1-6 frames in flight, frame batcher on and off --
against the oracle composition: every decoder an independent time de-interleaver + decode over exactly the CIFs it was handed.

    python tools/fuzz_mirror_lifecycle.py FIRST_SEED LAST_SEED            # FUZZ_REAL=1 on a GPU box: the same through libdabgpu.so
"""
import os, sys, subprocess, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import oracle as O, stream_model as SM
import test_mirror_host_logic as T
import test_mirror_lifecycle as L
O.build()
W = tempfile.mkdtemp(prefix="fuzz_lifecycle_")
objs = []
for src in ([] if os.environ.get("FUZZ_REAL") == "1" else T.ORACLE_SRCS):
    o = os.path.join(W, src + ".o")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-mavx2", "-mbmi2", "-mfma", "-c", os.path.join(T.ORACLE, src), "-o", o], check=True)
    objs.append(o)
REAL = os.environ.get("FUZZ_REAL") == "1"          # FUZZ_REAL=1 (on a GPU box): the driver built by __graft_entry__.build() on libdabgpu.so instead
EXE = os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver") if REAL else os.path.join(W, "mirror_lifecycle_fake")
if REAL:
    os.environ["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")
else:
  subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + T.HOST, "-I" + os.path.join(ROOT, "include"), "-I" + T.CSRC, "-I" + T.ORACLE,
                  os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
                  os.path.join(T.CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(T.HOST, s) for s in T.MIRROR_SRCS] + objs + ["-lm", "-o", EXE], check=True)
subs = [O.subchannel(v[0], v[1], eep_level=v[2], eep_type=v[3]) for v in L.SUBS.values()]
caps = {}
for name, c in {"a": dict(n_frames=26, seed=21, dropouts=((6, -0.03, 14000), (14, -0.03, 60000), (19, 0.2, 30000))),
                "b": dict(n_frames=26, seed=22, dropouts=((9, -0.02, 9000), (20, 0.3, 100000)))}.items():
    u8, _ = SM.make_offair_like_capture(O, c["n_frames"], subs, seed=c["seed"], dropouts=c["dropouts"])
    iq = O.iq_convert(u8, 0).view(np.complex64)
    iq.tofile(os.path.join(W, name + ".c32"))
    caps[name] = iq
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    which = "ab"[seed % 2]
    iq = caps[which]
    rng = np.random.default_rng(5000 + seed)
    lo, hi = [(100, 600000), (2000, 30000), (50000, 400000)][(seed // 2) % 3]
    schedule, total = [], 0
    while total < iq.size:
        n = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        if rng.random() < 0.03 and total > 0: n = -n
        schedule.append(n); total += abs(n)
    open(os.path.join(W, "schedule.txt"), "w").write("\n".join(map(str, schedule)) + "\n")
    model = SM.StreamModel(O); pos = 0
    for v in schedule:
        if pos >= iq.size: break
        if v < 0: model.reset()
        model.process(iq[pos:pos + abs(v)]); pos += abs(v)
    frames = [f["bits"] for f in model.out_frames]
    script = L.random_script(seed, n_frames=max(len(frames), 1))
    lines = [f"{fr} add {i} {L.SUBS[i][0]} {L.SUBS[i][1]} {L.SUBS[i][2]} {L.SUBS[i][3]}" if op == "add" else f"{fr} {op} {i}" for fr, op, i in script]
    open(os.path.join(W, "script.txt"), "w").write("\n".join(lines) + "\n")
    out = os.path.join(W, "out"); shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
    batch, depth = str(seed % 5 != 0 and 1 or 0), str(1 + seed % 6)
    res = subprocess.run([EXE, os.path.join(W, which + ".c32"), out, "65536", os.path.join(W, "script.txt")], capture_output=True, text=True,
                         env=dict(os.environ, DABGPU_HARNESS_SCHEDULE=os.path.join(W, "schedule.txt"), DABGPU_MIRROR_BATCH=batch, DABGPU_MIRROR_DEPTH=depth))
    ok = res.returncode == 0 and f"frames={len(frames)} " in res.stdout
    why = "" if ok else res.stdout.strip()[-150:] + res.stderr[-200:]
    if ok:
        fibs, msc = L.expected(O, frames, script)
        got = open(os.path.join(out, "fibs.bin"), "rb").read() if os.path.exists(os.path.join(out, "fibs.bin")) else b""
        if got != fibs: ok, why = False, "FIBs differ"
        for ident, exp in msc.items():
            if ok and open(os.path.join(out, f"msc_{ident}.bin"), "rb").read() != exp: ok, why = False, f"decoder {ident} differs"
    print(seed, which, len(frames), model.frames_desync, "batch" + batch, "depth" + depth, len(script), res.stdout.strip()[res.stdout.find("fib_groups"):][:110] if ok else "", "OK" if ok else "MISMATCH " + why, flush=True)
    bad += not ok
    if not ok:
        shutil.copy(os.path.join(W, "schedule.txt"), f"lifecycle_bad_{seed}_schedule.txt"); shutil.copy(os.path.join(W, "script.txt"), f"lifecycle_bad_{seed}_script.txt")
print("bad", bad)
shutil.rmtree(W, ignore_errors=True)
sys.exit(1 if bad else 0)
