"""The reference's own callers of the hot path, EXECUTED over the mirror classes on a FIG-valid synthetic ensemble (round 5 only linked them).

In the build container (skipped where /root/reference is absent): tests/ref_overlay.py compiles, inside an overlay of the reference tree,
  the reference's raw_u8 reader + OFDM_Block (examples/app_helpers/app_iq_readers.h, app_ofdm_blocks.h:25-58) + ThreadedRingBuffer,
  BasicFICRunner with the FIG processor / handler / database updater (src/basic_radio/basic_fic_runner.cpp:20-49, src/dab/fic/fig_processor.cpp:94),
  and tests/cpp/ref_callers_driver.cpp (BasicRadio::Process / UpdateAfterProcessing restated, basic_radio.cpp:41-65,83-154: BasicRadio itself needs faad2 / mpg123)
over the mirror sources and the oracle-backed C ABI, and runs them on the capture of tests/fig_ensemble.py.  Checked against the GENERATOR:
every FIB of every received frame reaches the FIG parser, the reference's database holds exactly the generator's sub-channels (start, length,
protection), services and components, every decoder is created in the frame the carousel completes its entry, and every decoder returns the
transmitted bytes of exactly the CIFs of its lifetime.  Checked against the committed fixture tests/golden/fig_ensemble.npz byte for byte (the
fixture is what -m gpu replays on the device: tests/test_gpu_fig_ensemble.py).

Everywhere (no reference needed): the fixture agrees with the generator regenerated from its seed."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FIXTURE = os.path.join(ROOT, "tests", "golden", "fig_ensemble.npz")


@pytest.fixture(scope="module")
def truth():
    import fig_ensemble as FE
    import oracle as O
    import stream_model as SM
    O.build()
    u8, t = FE.make_capture(O, SM)
    t["u8"] = u8
    return t


def check_against_generator(FE, truth, n_out, fibs, database, created, msc):
    """fibs [(frame, bytes)], database text, created [dict], msc {id: [(frame, cif, bytes)]}"""
    import dabfig
    desc, car = truth["desc"], truth["carousel"]
    # FIBs: a frame's twelve FIBs are received or (the first frame after acquisition: the fine frequency loop has not settled) none of them
    per_frame = {}
    for f, b in fibs:
        per_frame.setdefault(f, []).append(b)
    received = set(per_frame)
    assert all(len(v) == 12 for v in per_frame.values()), {f: len(v) for f, v in per_frame.items()}
    assert received >= set(range(1, n_out)), sorted(received)
    want = [x for x in FE.transmitted_fibs(truth, n_out) if x[0] in received]
    assert fibs == want, "the FIBs handed to the reference's FIG processor are not the transmitted ones"
    # the reference's database == the generator's description
    assert database.splitlines() == dabfig.expected_database(desc)
    # lazy creation: the frame in whose UpdateAfterProcessing each decoder appears, and with which parameters
    by_id = {s["id"]: s for s in desc["subchannels"]}
    want_created = {}
    for k, s in enumerate(desc["subchannels"]):
        f = car.decoder_after_frame(k, received)
        if f is not None and f < n_out:
            want_created[s["id"]] = f
    assert {c["id"]: c["frame"] for c in created} == want_created
    assert len(set(want_created.values())) >= 4, "the decoders must appear at different frames"
    assert by_id[desc["subchannels"][FE.STREAM_DATA_SUB]["id"]]["id"] not in want_created and desc["subchannels"][FE.ORPHAN_SUB]["id"] not in want_created
    for c in created:
        s = by_id[c["id"]]
        assert (c["start"], c["length"], c["is_uep"]) == (s["start"], s["length"], s["is_uep"]), c
        if s["is_uep"]:
            assert c["uep_index"] == s["uep_index"] and c["kind"] == "dab"
        else:
            assert (c["eep_level"], c["eep_type"]) == (s["eep_level"], s["eep_type"]) and c["kind"] == ("data_packet" if s["fec"] is not None else "dab_plus")
        assert c["fec"] == (255 if s["fec"] is None else s["fec"])
        # the decoder sees the frames after the one it was created in; its bytes are the transmitted ones
        got = msc[c["id"]]
        want = FE.transmitted_bytes(truth, s["index"], c["frame"] + 1, n_out)
        assert [(f, cif, len(b)) for f, cif, b in got] == [(f, cif, len(b)) for f, cif, b in want], c
        assert got == want, "sub-channel id %d: decoded bytes differ from the transmitted payload" % c["id"]
        assert sum(1 for _, _, b in got if b) >= 1


def load_fixture():
    z = np.load(FIXTURE)
    keys = ["frame", "id", "start", "length", "is_uep", "uep_index", "eep_level", "eep_type", "fec"]
    created = [dict(zip(keys, (int(v) for v in row)), kind=str(k)) for row, k in zip(z["created"], z["created_kind"])]
    msc = {}
    for c in created:
        idx, blob, pos, rec = z["msc_%d_index" % c["id"]], z["msc_%d_bytes" % c["id"]].tobytes(), 0, []
        for f, cif, n in idx:
            rec.append((int(f), int(cif), blob[pos:pos + int(n)]))
            pos += int(n)
        msc[c["id"]] = rec
    fibs = [(int(f), b.tobytes()) for f, b in zip(z["fib_frames"], z["fib_bytes"])]
    return dict(z=z, created=created, msc=msc, fibs=fibs, database=str(z["database"]), n_out=int(z["n_out_frames"]))


def test_fixture_agrees_with_the_generator(truth):
    """runs anywhere: the committed fixture belongs to the capture its seed regenerates, and what it holds is what was transmitted"""
    import fig_ensemble as FE
    fx = load_fixture()
    z = fx["z"]
    assert int(z["seed"]) == FE.SEED and int(z["n_frames"]) == FE.N_FRAMES and int(z["block"]) == FE.BLOCK
    assert str(z["capture_sha256"]) == truth["sha256"], "the capture regenerated here is not the one the fixture was made from"
    check_against_generator(FE, truth, fx["n_out"], fx["fibs"], fx["database"], fx["created"], fx["msc"])


def test_reference_callers_run_over_the_mirror(truth, tmp_path):
    import ref_overlay as RO
    ok, why = RO.available()
    if not ok:
        pytest.skip(why)
    import fig_ensemble as FE
    import make_golden_fig_ensemble as MG
    r = MG.run_driver(str(tmp_path))
    assert r["truth"]["sha256"] == truth["sha256"]
    d = MG.collect(r)
    n_out = int(d["n_out_frames"])
    assert n_out == FE.N_FRAMES - 1 and "frames=%d read=%d desync=0" % (n_out, n_out) in r["stdout"], r["stdout"]
    created = FE.read_created(os.path.join(r["out"], "created.txt"))
    msc = {c["id"]: FE.read_msc(os.path.join(r["out"], "msc_%d.bin" % c["id"])) for c in created}
    check_against_generator(FE, truth, n_out, FE.read_fibs(os.path.join(r["out"], "fibs.bin")), open(os.path.join(r["out"], "database.txt")).read(), created, msc)
    # the OFDM_Block's output stream == the serial oracle state machine over the reference reader's samples
    import oracle as O
    import stream_model as SM
    iq = O.iq_convert(truth["u8"], 0).view(np.complex64)
    model = SM.StreamModel(O)
    for k in range(0, iq.size, FE.BLOCK):
        model.process(iq[k:k + FE.BLOCK])
    assert len(model.out_frames) == n_out
    assert [hashlib.sha256(f["bits"].tobytes()).hexdigest() for f in model.out_frames] == [str(s) for s in d["frame_bits_sha256"]]
    # ... and the committed fixture, byte for byte
    z = np.load(FIXTURE)
    assert sorted(z.files) == sorted(d.keys())
    for k in d:
        assert np.array_equal(np.asarray(d[k]), z[k]), "fixture field %s is stale: run tests/golden/make_golden_fig_ensemble.py" % k
