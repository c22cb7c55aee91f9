"""-m gpu: DAB_Viterbi_Decoder (the C++ mirror class over the C ABI) is as general as the reference's class
(src/dab/algorithms/dab_viterbi_decoder.cpp:109-181): ANY puncture vector, ANY requested_output_symbols, any number of update() calls,
non-zero start and end states, a chainback shorter than the decoded length.  Scripted random call sequences run through
tests/cpp/viterbi_harness and through the oracle's Viterbi class; every return value of every call -- symbols consumed,
get_current_decoded_bit(), path error, decoded bytes -- must be identical, for both tie rules."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "viterbi_harness")


def make_cases(oracle, rng, n_cases):
    """random call sequences over noisy transmissions of random messages (so that the survivor is well defined), plus pure noise"""
    cases = []
    for c in range(n_cases):
        n_bits = int(rng.integers(1, 60)) * 8 if c % 7 else int(rng.integers(1, 400)) * 8
        msg = rng.integers(0, 256, n_bits // 8, dtype=np.uint8)
        mother = oracle.conv_encode(msg)                                     # 4 * (n_bits + 6) bits 0/1, zero tail
        soft = (oracle.soft_from_bits(mother).astype(np.int16))
        if c % 5 == 4:
            soft = rng.integers(-127, 128, soft.size).astype(np.int16)      # noise only: ties and wrap-around territory
        else:
            soft = np.clip(soft + rng.normal(0, 60 + 40 * (c % 3), soft.size), -127, 127).astype(np.int16)
        soft = soft.astype(np.int8)
        total_steps = n_bits + 6
        # cut the steps into updates with random puncture vectors
        updates, step, pos_m = [], 0, 0
        while step < total_steps:
            n_code = int(rng.integers(1, 10))
            code = rng.integers(0, 5, n_code).astype(np.uint8)
            if c % 4 == 0:
                code = np.maximum(code, 1)
            if c % 11 == 3:
                code = oracle.puncture_code(int(rng.integers(1, 25)))
            steps_here = int(min(total_steps - step, rng.integers(1, 1 + max(1, total_steps // 3))))
            kept = []
            for g in range(steps_here):
                k = int(code[g % len(code)])
                kept.append(soft[pos_m:pos_m + k])
                pos_m += 4
            punct = np.concatenate(kept) if kept else np.zeros(0, np.int8)
            extra = rng.integers(-127, 128, int(rng.integers(0, 9))).astype(np.int8)       # the span may be longer than what is consumed
            updates.append((code, 4 * steps_here, np.concatenate([punct, extra])))
            step += steps_here
        if c % 13 == 6 and updates:                                         # one update that runs out of symbols: consumes nothing, decodes nothing
            code, req, p = updates[-1]
            if int(code.sum()) > 0 and p.size > 1:
                need = sum(int(code[g % len(code)]) for g in range(req // 4))
                if need > 1:
                    updates.insert(len(updates) - 1, (code, req, p[:need - 1].copy()))
        start = int(rng.integers(0, 64)) if c % 3 == 1 else 0
        end = int(rng.integers(0, 64)) if c % 3 == 2 else 0
        n_out = (total_steps - 6) // 8
        if c % 6 == 5 and n_out > 2:
            n_out = int(rng.integers(1, n_out))                              # chainback shorter than what was decoded
        if c % 17 == 16:
            n_out = n_out + 3                                                # ... and longer: refused (the reference reads stale decision words)
        cases.append(dict(start=start, end=end, n_out=n_out, updates=updates))
    return cases


def write_script(path, cases):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(cases)))
        for cs in cases:
            f.write(struct.pack("<IIII", cs["start"], cs["end"], cs["n_out"], len(cs["updates"])))
            for code, req, p in cs["updates"]:
                f.write(struct.pack("<I", len(code))); f.write(code.tobytes())
                f.write(struct.pack("<II", req, p.size)); f.write(p.tobytes())


@pytest.mark.parametrize("tie_rule", [0, 1])
def test_general_update_sequences_match_the_oracle_class(oracle, tmp_path, tie_rule):
    if not os.path.exists(HARNESS):
        import __graft_entry__ as g
        g.build()
    rng = np.random.default_rng(500 + tie_rule)
    cases = make_cases(oracle, rng, 120)
    script, out = tmp_path / "script.bin", tmp_path / "out.bin"
    write_script(script, cases)
    env = dict(os.environ, DABGPU_TIE_RULE=str(tie_rule))
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([HARNESS, str(script), str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    blob = open(out, "rb").read()
    pos = 0
    n_short = n_refused = n_starved = 0
    for ci, cs in enumerate(cases):
        v = oracle.Viterbi(1 << 16, tie_rule)
        v.reset(cs["start"])
        steps = 0
        for code, req, p in cs["updates"]:
            used_o = v.update(p, code, req)
            used_g, = struct.unpack_from("<Q", blob, pos); pos += 8
            assert used_g == used_o, (ci, "consumed")
            if p.size < sum(int(code[g % len(code)]) for g in range(req // 4)):      # ran out of symbols: nothing consumed, nothing decoded
                assert used_o == 0
                n_starved += 1
            else:
                steps += req // 4
        dec_g, status, err_g = struct.unpack_from("<QIQ", blob, pos); pos += 20
        assert dec_g == v.decoded_bits() == steps, (ci, "current decoded bit")
        got = np.frombuffer(blob, np.uint8, cs["n_out"], pos); pos += cs["n_out"]
        if cs["n_out"] * 8 + 6 > steps:
            assert status == 1 and (got == 0xEE).all(), (ci, "a trace-back beyond the decoded steps must be refused, bytes untouched")
            n_refused += 1
            continue
        assert status == 0, (ci, res.stderr[-500:])
        exp, err_o = v.chainback(cs["n_out"], cs["end"])
        assert np.array_equal(got, exp), (ci, "bytes", cs["n_out"], steps)
        assert err_g == err_o, (ci, "path error")
        n_short += cs["n_out"] * 8 + 6 < steps
    assert pos == len(blob)
    assert n_short >= 5 and n_refused >= 3 and n_starved >= 1           # the script exercised the unusual shapes
