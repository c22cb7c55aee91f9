"""-m gpu checks at BASELINE.json's sizes through size-independent properties (the oracle only samples at these sizes):
configs[1] (1024 frames of the demodulator: hard bits == transmitted bits for every frame, soft bits == oracle on sampled
frames) through bench.py itself, and configs[2]/[3] (1024 ensembles x 18 sub-channels: every FIB CRC passes, every decoded
byte equals the transmitted payload) through tools/bench_decode.py -- the same programs the measurements come from."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_json(args):
    res = subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_contract_and_full_size_demod_check():
    d = run_json(["bench.py", "--steps", "20", "--warmup", "3", "--prewarm-ms", "50", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "dab_mode1_frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 20
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"] == "f32"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["config"]["frames_per_gpu_per_step"] == 1024
    c = d["check"]
    assert c["frames_checked"] == 1024 and c["hard_bit_errors_vs_transmitted"] == 0 and c["soft_bit_mismatches_vs_oracle_3_frames"] == 0
    assert d["value"] > 1.0e6, "an MI355X demodulates far more than a million frames per second"


def test_full_pipeline_at_1024_ensembles():
    d = run_json(["tools/bench_decode.py", "--ensembles", "1024", "--steps", "2"])
    c = d["check"]
    assert c["fib_crc_pass"] == c["fib_crc_expected"] == 12288
    assert c["fib_bytes_equal_transmitted"] is True and c["msc_bytes_equal_transmitted"] is True
