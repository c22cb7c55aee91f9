"""-m gpu checks at BASELINE.json's sizes through size-independent properties (the oracle only samples at these sizes), run through the
same programs the measurements come from:
configs[1] (1024 frames of the demodulator: hard bits == transmitted bits for every frame, soft bits == oracle on sampled frames) and
configs[2]/[3] (4096 ensembles x 18 sub-channels, two frames in flight: every FIB CRC passes, every decoded byte equals the transmitted
payload) through bench.py's default line; configs[4] per GPU (8192 ensembles, bench.py --workload full) with the same checks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_json(args):
    res = subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, timeout=1200)
    assert res.returncode == 0, (res.stderr[-3000:], res.stdout[-500:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_contract_and_full_size_checks():
    d = run_json(["bench.py", "--steps", "20", "--warmup", "3", "--prewarm-ms", "50", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "extra"):
        assert key in d, key
    assert d["metric"] == "dab_mode1_frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 20
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"] == "f32"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02, "the dominant kernel cannot take longer than the step it is part of"
    assert d["config"]["frames_per_gpu_per_step"] == 1024
    c = d["check"]
    assert c["frames_checked"] == 1024 and c["hard_bit_errors_vs_transmitted"] == 0 and c["soft_bit_mismatches_vs_oracle_3_frames"] == 0
    assert d["value"] > 1.0e6, "an MI355X demodulates far more than a million frames per second"
    # configs[2] / configs[3] at BASELINE's 4096 ensembles
    c2, c3 = d["extra"]["configs2"], d["extra"]["configs3"]
    assert c2["frames"] == 4096 and c3["ensembles"] == 4096 and c3["frames_in_flight"] == 2
    k2, k3 = c2["check"], c3["check"]
    assert k2["fib_crc_pass"] == k2["fib_crc_expected"] and k2["fib_bytes_equal_transmitted"] is True
    assert k3["fib_crc_pass"] == k3["fib_crc_expected"] == 2 * 4096 * 12
    assert k3["fib_bytes_equal_transmitted"] is True and k3["msc_bytes_equal_transmitted"] is True and k3["ensembles_checked"] == 4096
    for blk in c2["roofline"] + c3["roofline"]:
        assert 0.0 < blk["frac"] < 1.0 and blk["bound"] in ("hbm", "valu_issue")


def test_full_workload_of_configs4_per_gpu():
    d = run_json(["bench.py", "--workload", "full", "--steps", "4", "--warmup", "1", "--prewarm-ms", "0", "--no-cpu-baseline"])
    assert d["config"]["ensembles_per_gpu"] == 8192 and d["n_gpus"] == 1 and d["scaling"] == "weak"
    c = d["check"]
    assert c["ensembles_checked"] == 8192 and c["fib_crc_pass"] == c["fib_crc_expected"]
    assert c["fib_bytes_equal_transmitted"] is True and c["msc_bytes_equal_transmitted"] is True
    assert d["value"] > 2.0e5


def test_stage_by_stage_tool_at_1024_ensembles():
    d = run_json(["tools/bench_decode.py", "--ensembles", "1024", "--steps", "2"])
    c = d["check"]
    assert c["fib_crc_pass"] == c["fib_crc_expected"] == 12288
    assert c["fib_bytes_equal_transmitted"] is True and c["msc_bytes_equal_transmitted"] is True
    assert d["config4_full"]["two_frames_in_flight_outputs_ok"] is True
