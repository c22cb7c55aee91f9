"""Generates tests/golden/fig_ensemble.npz: what the REFERENCE's own callers produce when they are executed over the mirror classes on the
FIG-valid synthetic ensemble of tests/fig_ensemble.py (build container only; needs /root/reference).

    python tests/golden/make_golden_fig_ensemble.py

Pipeline executed (tests/cpp/ref_callers_driver.cpp, built by tests/ref_overlay.py inside an overlay of the reference tree):
  the reference's raw_u8 reader (examples/app_helpers/app_iq_readers.h) -> the reference's OFDM_Block::run (app_ofdm_blocks.h:45-58) over the
  mirror OFDM_Demod -> the reference's ThreadedRingBuffer -> per frame: the reference's BasicFICRunner::Process (basic_fic_runner.cpp:34-49: mirror
  FIC_Decoder -> the reference's FIG_Processor -> Radio_FIG_Handler -> DAB_Database_Updater) and the MSC decoders created lazily from the
  reference's database as basic_radio.cpp:83-154 does.
Below the classes sits the oracle-backed C ABI (tests/cpp/fake_dabgpu_oracle.cpp): THE NUMBERS IN THIS FIXTURE ARE THE ORACLE'S (FFT rounding and
the ACS core remain restated); what the fixture pins is the reference's control flow around them -- which FIBs reach the FIG
parser, what the database holds, at which frame each decoder appears and which CIFs it is handed -- and -m gpu replays it on the device
(tests/test_gpu_fig_ensemble.py).  Stored: capture seed / SHA-256 (the capture is regenerated, not committed), SHA-256 of every frame's soft
bits, every CRC-valid FIB, the database dump, the creation records, every DecodeCIF result."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run_driver(workdir, exe=None):
    """-> dict of everything the driver wrote (+ truth); workdir gets capture.u8 and out/"""
    import fig_ensemble as FE
    import oracle as O
    import ref_overlay as RO
    import stream_model as SM
    O.build()
    u8, truth = FE.make_capture(O, SM)
    cap = os.path.join(workdir, "capture.u8")
    u8.tofile(cap)
    if exe is None:
        exe = RO.Overlay(os.path.join(workdir, "overlay")).callers_driver("oracle")
    out = os.path.join(workdir, "out")
    os.makedirs(out, exist_ok=True)
    res = subprocess.run([exe, cap, out, str(FE.BLOCK)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return dict(FE=FE, O=O, truth=truth, out=out, stdout=res.stdout, exe=exe)


def collect(r):
    FE, out = r["FE"], r["out"]
    bits = np.fromfile(os.path.join(out, "frame_bits.bin"), dtype=np.int8).reshape(-1, 230400)
    fibs = FE.read_fibs(os.path.join(out, "fibs.bin"))
    created = FE.read_created(os.path.join(out, "created.txt"))
    d = dict(seed=np.int64(FE.SEED), n_frames=np.int64(FE.N_FRAMES), block=np.int64(FE.BLOCK), capture_sha256=np.array(r["truth"]["sha256"]),
             n_out_frames=np.int64(bits.shape[0]), frame_bits_sha256=np.array([hashlib.sha256(b.tobytes()).hexdigest() for b in bits]),
             fib_frames=np.array([f for f, _ in fibs], np.int32), fib_bytes=np.frombuffer(b"".join(b for _, b in fibs), np.uint8).reshape(-1, 30),
             database=np.array(open(os.path.join(out, "database.txt")).read()),
             created=np.array([[c["frame"], c["id"], c["start"], c["length"], c["is_uep"], c["uep_index"], c["eep_level"], c["eep_type"], c["fec"]] for c in created], np.int32),
             created_kind=np.array([c["kind"] for c in created]), stdout=np.array(r["stdout"].strip()))
    for c in created:
        rec = FE.read_msc(os.path.join(out, "msc_%d.bin" % c["id"]))
        d["msc_%d_index" % c["id"]] = np.array([[f, cif, len(b)] for f, cif, b in rec], np.int32).reshape(-1, 3)
        d["msc_%d_bytes" % c["id"]] = np.frombuffer(b"".join(b for _, _, b in rec), np.uint8)
    return d


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as wd:
        r = run_driver(wd)
        d = collect(r)
        np.savez_compressed(os.path.join(HERE, "fig_ensemble.npz"), **d)
        print(r["stdout"].strip())
        print("frames", int(d["n_out_frames"]), "FIBs", len(d["fib_frames"]), "decoders", len(d["created"]), "->", os.path.join(HERE, "fig_ensemble.npz"))
