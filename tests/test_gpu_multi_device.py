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


def test_soak_at_the_fan_out_of_an_eight_gpu_node_on_one_device():
    """No 8-GPU node exists for the tests: eight workers -- eight host threads, sixteen contexts, sixteen streams, their own multiplexes --
    share GPU 0 for 60 steps with two frames in flight each (SURVEY 8e's host structure at its real fan-out).  Every decoded byte of
    every worker equals what it transmitted, the workers decode different bytes, and the device's free memory after step 10 equals the
    free memory after step 60: nothing in the library grows per call.  (Scaling itself stays unmeasured.)"""
    out = run("--devices", "0,0,0,0,0,0,0,0", "--ensembles", "512", "--steps", "60", "--distinct", "16", "--mem-probe-step", "10")
    assert out["all_outputs_equal_transmitted"] and out["workers"] == 8 and out["frames_per_s"] > 0
    assert len({w["digest"] for w in out["per_worker"]}) == 8
    for w in out["per_worker"]:
        assert w["ok"] and w["fib_groups_wrong"] == 0 and w["msc_cifs_wrong"] == 0 and w["fib_crc_pass"] == w["fib_crc_expected"] == 2 * 512 * 12
        assert w["symbols_per_block"] in (25, 38, 75)
        assert w["device_free_bytes_at_probe"] > 0
        # one device, eight observers: each sees the same pool; allow nothing beyond the allocator's 2 MiB granule
        assert abs(w["device_free_bytes_at_end"] - w["device_free_bytes_at_probe"]) <= (2 << 20), w
