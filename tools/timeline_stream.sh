export TMPDIR=/tmp
rm -rf gpurun_out/pk
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pk -o t -- python3 tools/bench_stream.py --streams 1024 --block-frames 4 > gpurun_out/pk_stdout.log 2>&1
f=$(find gpurun_out/pk -name '*kernel_trace.csv' | head -1)
n=$(python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open('$f')) if 'dabgpu' in r['Kernel_Name']]
print(len(rows))")
echo rows $n
python3 tools/ktimeline.py $f $((n-75)) 75
rm -rf gpurun_out/pk
