"""bench.py's own multi-rank launcher (`python bench.py --gpus N` started plainly): spawns the ranks as a child process, rendezvous on
127.0.0.1, shards the units, barrier + max over ranks, ONE JSON line from rank 0, child failures become a non-zero exit status.
CPU only (gloo, --dry-run: no kernels) -- the GPU path differs only in the backend (nccl = RCCL) and in the step body."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "bench.py", "--dry-run", "--steps", "5", "--warmup", "1"] + extra, cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)


def test_plain_launch_spawns_two_ranks_and_prints_one_line():
    res = run(["--gpus", "2"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 5
    assert d["config"]["units_per_rank"] == 1024 and d["config"]["units_covered"] == 2048      # weak scaling: every rank its own 1024 frames
    assert d["ms_per_step"] >= 2.0                                                              # max over ranks: rank 1 sleeps 2 ms per step
    # every rank's own step time travels with the line (a straggler shows): rank 0 sleeps 1 ms per step, rank 1 two
    assert len(d["ms_per_step_per_rank"]) == 2 and d["ms_per_step_per_rank"][1] > d["ms_per_step_per_rank"][0] >= 1.0
    assert max(d["ms_per_step_per_rank"]) <= d["ms_per_step"] * 1.001


def test_full_workload_shards_8192_ensembles_per_rank():
    res = run(["--gpus", "2", "--workload", "full"])
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["units_per_rank"] == 8192 and d["config"]["units_covered"] == 16384


def test_failing_rank_gives_nonzero_exit_status():
    res = run(["--gpus", "2", "--dry-run-fail-rank", "1"])
    assert res.returncode != 0
