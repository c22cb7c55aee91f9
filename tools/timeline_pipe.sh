# development: kernel timeline of the configs[4] step (sync + demod + FIC + MSC, two frames in flight) at 4096 ensembles
export TMPDIR=/tmp
rm -rf gpurun_out/pk
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pk -o t -- python3 bench.py --workload full --ensembles ${1:-4096} --steps 6 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-check > gpurun_out/pk_stdout.log 2>&1
f=$(find gpurun_out/pk -name '*kernel_trace.csv' | head -1)
n=$(python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open('$f')) if 'dabgpu' in r['Kernel_Name']]
print(len(rows))")
echo rows $n
python3 tools/ktimeline.py $f $((n-${2:-120})) 70
rm -rf gpurun_out/pk
