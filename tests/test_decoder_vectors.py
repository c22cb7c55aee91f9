"""tests/golden/decoder_vectors.npz: the REFERENCE's FIC_Decoder::DecodeFIBGroup (src/dab/fic/fic_decoder.cpp:53-117), MSC_Decoder::DecodeCIF / DecodeEEP /
DecodeUEP (src/dab/msc/msc_decoder.cpp:46-154) and CIF_Deinterleaver, compiled in place and EXECUTED over 48 FIB groups and 20 CIFs of every protection
profile its tables hold (EEP 1-A..4-A incl. the 2-A n = 1 special case, 1-B..4-B, all 64 UEP rows incl. the exchanged size fields of rows 33 / 34), under
both core models.  Label of the vectors: reference control code over the RESTATED Viterbi core (tests/golden/make_golden_decoders.py): they pin
SURVEY 8 rows a20-a22 -- slicing, segment plans, padding, the order of the update() calls, chain-back length, descrambler, CRC -- not the core.

CPU (here): the oracle's composition (fic_decode_group; Deinterleaver + msc_decode_logical) equals the vectors; where oracle/_ref/libdab_ref_decoders.so
exists, the library run live equals the vectors too.  -m gpu: tests/test_gpu_decoder_vectors.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
FIXTURE = os.path.join(ROOT, "tests", "golden", "decoder_vectors.npz")


def load():
    """-> dict(z, cases [dict], fic {model: (n_valid, fibs)}, msc {model: [case][cif] bytes})"""
    z = np.load(FIXTURE)
    keys = ["mux", "start", "length", "is_uep", "uep_index", "eep_level", "eep_type", "setting"]
    cases = [dict(zip(keys, (int(v) for v in row)), name=str(n), index=k) for k, (row, n) in enumerate(zip(z["case_table"], z["case_names"]))]
    blob0 = z["msc_bytes_scalar"]
    blobs = {0: blob0.tobytes(), 1: (blob0 ^ z["msc_bytes_simd_xor_scalar"]).tobytes()}
    msc = {}
    for m in (0, 1):
        pos, per_case = 0, []
        for row in z["msc_lengths"]:
            rec = []
            for n in row:
                rec.append(blobs[m][pos:pos + int(n)])
                pos += int(n)
            per_case.append(rec)
        assert pos == len(blobs[m])
        msc[m] = per_case
    fic = {0: (z["fic_n_valid_scalar"], z["fic_fibs_scalar"]), 1: (z["fic_n_valid_simd"], z["fic_fibs_scalar"] ^ z["fic_fibs_simd_xor_scalar"])}
    return dict(z=z, cases=cases, fic=fic, msc=msc)


@pytest.fixture(scope="module")
def inputs(oracle):
    import decoder_vectors as DV
    cs, cifs, payload = DV.msc_multiplexes(oracle)
    soft_fic, fib_data = DV.fic_groups(oracle)
    return dict(DV=DV, cases=cs, cifs=cifs, payload=payload, soft_fic=soft_fic, fib_data=fib_data)


def test_inputs_regenerate_and_the_vectors_make_sense(oracle, inputs):
    fx = load()
    z, DV = fx["z"], inputs["DV"]
    assert "RESTATED Viterbi core" in str(z["label"])
    assert DV.sha(inputs["soft_fic"]) == str(z["fic_inputs_sha256"]) and DV.sha(inputs["cifs"]) == str(z["msc_inputs_sha256"]), "regenerated inputs differ"
    assert [c["name"] for c in inputs["cases"]] == [c["name"] for c in fx["cases"]]
    for a, b in zip(inputs["cases"], fx["cases"]):
        assert all(a[k] == b[k] for k in ("mux", "start", "length", "is_uep", "uep_index", "eep_level", "eep_type", "setting"))
    names = [c["name"] for c in fx["cases"]]
    assert sum(n.startswith("uep_") for n in names) == 64 and sum(n.startswith("eep_") for n in names) == 11 and "eep_2A_n1" in names
    # nothing comes out before the 16th CIF, then every CIF gives the profile's byte count (cif_deinterleaver.cpp:40-42, msc_decoder.cpp:53-63)
    lens = z["msc_lengths"]
    assert (lens[:, :15] == 0).all() and (lens[:, 15:] > 0).all() and all(len(set(r[15:])) == 1 for r in lens)
    # the clean setting decodes to what was transmitted (logical frame of CIF t is the payload of CIF t - 15), under both models; the noisy ones need not
    n_clean = 0
    for c, pay in zip(fx["cases"], inputs["payload"]):
        if c["setting"] == 0 and pay is not None:
            for m in (0, 1):
                assert all(fx["msc"][m][c["index"]][t] == pay[t - 15].tobytes() for t in range(15, 20)), c["name"]
            n_clean += 1
    assert n_clean >= 24
    # FIC: the clean groups deliver their three FIBs, the noise-only ones none; the models disagree somewhere (exact ties exist in these inputs)
    n0, f0 = fx["fic"][0]
    for g in range(len(n0)):
        if g % 8 == 0:
            assert n0[g] == 3 and f0[g].tobytes() == inputs["fib_data"][g].tobytes()
        if g % 8 == 6:
            assert n0[g] == 0
    assert np.count_nonzero(z["msc_bytes_simd_xor_scalar"]) > 100


@pytest.mark.parametrize("model", [0, 1], ids=["scalar_core", "simd_core"])
def test_oracle_composition_equals_the_reference_control_code(oracle, inputs, model):
    fx, DV = load(), inputs["DV"]
    n, fibs = DV.oracle_fic(oracle, inputs["soft_fic"], model)
    assert np.array_equal(n, fx["fic"][model][0]) and np.array_equal(fibs, fx["fic"][model][1])
    for c in inputs["cases"]:
        got = DV.oracle_msc(oracle, c, inputs["cifs"][c["mux"]], model)
        want = fx["msc"][model][c["index"]]
        assert [b"" if g is None else g.tobytes() for g in got] == want, c["name"]


def test_reference_decoders_run_live_equal_the_fixture(oracle):
    R = oracle.ref_decoders()
    if R is None:
        pytest.skip("oracle/_ref/libdab_ref_decoders.so is only built where /root/reference exists")
    import make_golden_decoders as MG
    d = MG.make(oracle, R)
    z = np.load(FIXTURE)
    assert sorted(d.keys()) == sorted(z.files)
    for k in d:
        assert np.array_equal(np.asarray(d[k]), z[k]), "fixture field %s is stale: run tests/golden/make_golden_decoders.py" % k
