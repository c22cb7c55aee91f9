"""-m gpu: the C++ multi-device host (tests/cpp/multi_gpu_harness.cpp): one host thread, one set of contexts and streams per worker,
BASELINE configs[4]'s step per worker.  No node with several GPUs is available to the tests, so two workers share GPU 0
(`--devices 0,0`): every decoded byte of both must equal what they transmitted, and with identical ensembles both workers must
produce the digest of the single-worker run -- two contexts decoding side by side do not disturb each other."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "multi_gpu_harness")


def run(*args):
    if not os.path.exists(HARNESS):
        import __graft_entry__ as g
        g.build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([HARNESS, *args], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-2000:])
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_two_workers_on_one_gpu_equal_the_single_worker_run():
    one = run("--devices", "0", "--ensembles", "40", "--steps", "3", "--identical")
    two = run("--devices", "0,0", "--ensembles", "40", "--steps", "3", "--identical")
    assert one["all_outputs_equal_transmitted"] and two["all_outputs_equal_transmitted"]
    assert one["workers"] == 1 and two["workers"] == 2 and two["frames_per_s"] > 0
    d = one["per_worker"][0]["digest"]
    assert [w["digest"] for w in two["per_worker"]] == [d, d]
    for w in two["per_worker"]:
        assert w["fib_groups_wrong"] == 0 and w["msc_cifs_wrong"] == 0 and w["fib_crc_pass"] == w["fib_crc_expected"] == 2 * 40 * 12


def test_workers_with_their_own_ensembles_and_one_frame_in_flight():
    out = run("--devices", "0,0,0", "--ensembles", "17", "--steps", "2", "--distinct", "5", "--inflight", "1")
    assert out["all_outputs_equal_transmitted"] and out["workers"] == 3
    assert len({w["digest"] for w in out["per_worker"]}) == 3            # worker r starts at multiplex r: different bytes
