#!/bin/bash
# per-kernel times of a python command on the GPU box:  bash tools/prof_kernels.sh tools/bench_decode.py --mapping 2
export TMPDIR=/tmp
rm -rf gpurun_out/pk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -o t -- python3 "$@" > gpurun_out/pk_stdout.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pk/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "dabgpu" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
rm -rf gpurun_out/pk
