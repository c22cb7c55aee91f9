"""Replay of tests/golden/fig_ensemble.npz -- what the REFERENCE's callers produced when they were executed over the mirror classes in the build
container (tests/golden/make_golden_fig_ensemble.py: OFDM_Block, BasicFICRunner + FIG processor + database updater, decoders created lazily from
the reference's database, basic_radio.cpp:83-154) -- on the device, where no reference source exists:

  * the capture is regenerated from the fixture's seed (SHA-256 checked),
  * tests/cpp/mirror_lifecycle_driver (the mirror classes over libdabgpu.so) gets a script that creates each MSC_Decoder in the frame after the
    one the reference completed its database entry in, with the parameters the reference's database held,
  * every CRC-valid FIB and every DecodeCIF result must equal the fixture's, record for record; dabgpu_radio_cli's frame bits must hash to
    the fixture's frame hashes (= the OFDM_Block's output stream).
The same replay runs on the oracle-backed ABI as a CPU test (the host logic of the classes on this capture)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DRIVER = os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver")
CLI = os.path.join(ROOT, "dab-radio_amd", "host", "apps", "dabgpu_radio_cli")
LIB_ENV = {"LD_LIBRARY_PATH": os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")}


@pytest.fixture(scope="module")
def replay(tmp_path_factory):
    import fig_ensemble as FE
    import oracle as O
    import stream_model as SM
    import test_reference_callers_run as TR
    O.build()
    fx = TR.load_fixture()
    u8, truth = FE.make_capture(O, SM, seed=int(fx["z"]["seed"]), n_frames=int(fx["z"]["n_frames"]))
    assert truth["sha256"] == str(fx["z"]["capture_sha256"]), "the capture regenerated here is not the one the fixture was made from"
    d = tmp_path_factory.mktemp("fig_ensemble")
    u8.tofile(d / "capture.u8")
    O.iq_convert(u8, 0).view(np.complex64).tofile(d / "capture.c32")          # the reference reader's arithmetic (pinned: tests/test_io_formats.py)
    lines = ["0 fic 1"]
    for c in fx["created"]:
        lines.append("%d add %d %d %d %d %d %d %d" % (c["frame"] + 1, c["id"], c["start"], c["length"], c["eep_level"], c["eep_type"], c["is_uep"], c["uep_index"]))
    (d / "script.txt").write_text("\n".join(lines) + "\n")
    return dict(FE=FE, fx=fx, dir=d, block=int(fx["z"]["block"]))


def run_replay(exe, replay, out, env):
    FE, fx = replay["FE"], replay["fx"]
    out.mkdir()
    res = subprocess.run([exe, str(replay["dir"] / "capture.c32"), str(out), str(replay["block"]), str(replay["dir"] / "script.txt")], capture_output=True, text=True,
                         env=dict(os.environ, **env), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "frames=%d " % fx["n_out"] in res.stdout, res.stdout
    assert FE.read_fibs(str(out / "fibs.bin")) == fx["fibs"], "the FIBs differ from the ones the reference's FIG processor was handed"
    for c in fx["created"]:
        got = FE.read_msc(str(out / ("msc_%d.bin" % c["id"])))
        assert got == fx["msc"][c["id"]], "sub-channel id %d (%s): DecodeCIF results differ from the reference callers' run" % (c["id"], c["kind"])
    return res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("batch,depth", [("1", 3), ("0", 3), ("1", 1), ("1", 6)])
def test_replay_on_the_device(replay, tmp_path, batch, depth):
    if not os.path.exists(DRIVER):
        import __graft_entry__ as g
        g.build()
    text = run_replay(DRIVER, replay, tmp_path / "out", dict(LIB_ENV, DABGPU_MIRROR_BATCH=batch, DABGPU_MIRROR_DEPTH=str(depth)))
    if batch == "1" and depth <= 3:
        import re
        k = {m.group(1): int(m.group(2)) for m in re.finditer(r"(\w+)=(\d+)", text)}
        assert k["cifs_batched"] > 0 and k["fib_groups_batched"] > 0, k          # the frames' batched decodes really served the classes


@pytest.mark.gpu
def test_cli_frame_bits_are_the_ofdm_blocks_output(replay, tmp_path):
    if not os.path.exists(CLI):
        import __graft_entry__ as g
        g.build()
    fx = replay["fx"]
    out = tmp_path / "bits.bin"
    res = subprocess.run([CLI, "-i", str(replay["dir"] / "capture.u8"), "--configuration", "dab+ofdm", "--ofdm-input-mode", "raw_u8", "--ofdm-block-size", str(replay["block"]),
                          "--ofdm-enable-output", "--ofdm-output", str(out), "--radio-fib-output", str(tmp_path / "fibs.bin")],
                         capture_output=True, env=dict(os.environ, **LIB_ENV), timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    bits = np.fromfile(out, dtype=np.int8).reshape(-1, 230400)
    assert [hashlib.sha256(b.tobytes()).hexdigest() for b in bits] == [str(s) for s in fx["z"]["frame_bits_sha256"]]
    assert (tmp_path / "fibs.bin").read_bytes() == b"".join(b for _, b in fx["fibs"])


def test_replay_on_the_oracle_backed_abi(replay, tmp_path):
    """CPU: the same replay through the classes' host logic (tests/cpp/fake_dabgpu_oracle.cpp below them)"""
    import test_mirror_host_logic as T
    d = tmp_path / "build"
    d.mkdir()
    objs = []
    for src in T.ORACLE_SRCS:
        o = d / (src + ".o")
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-w", "-mavx2", "-mbmi2", "-mfma", "-c", os.path.join(T.ORACLE, src), "-o", str(o)],
                       check=True, timeout=600)
        objs.append(str(o))
    exe = d / "mirror_lifecycle_fake"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + T.HOST, "-I" + os.path.join(ROOT, "include"), "-I" + T.CSRC, "-I" + T.ORACLE,
                    os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_dabgpu_oracle.cpp"),
                    os.path.join(T.CSRC, "dabgpu_host_logic.cpp")] + [os.path.join(T.HOST, s) for s in T.MIRROR_SRCS] + objs + ["-lm", "-o", str(exe)],
                   check=True, timeout=900)
    run_replay(str(exe), replay, tmp_path / "out", dict(DABGPU_MIRROR_BATCH="1", DABGPU_MIRROR_DEPTH="3"))
