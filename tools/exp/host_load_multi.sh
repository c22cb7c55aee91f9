# experiment: several receivers in one process, threads pinned to the GPU's NUMA node (cores 64-127 on the pool's boxes), 3 repetitions
export TMPDIR=/tmp
D=/tmp/tmm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers 8 --frames 2400 --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_DRIVER_BENCH=1 LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
for rep in 1 2 3; do
for R in 1 2 4 8; do
  FILES=""; for k in $(seq 0 $((R-1))); do FILES="$FILES $D/rx$k.c32"; done
  out=$(taskset -c 64-127 ./tests/cpp/mirror_threads_driver 65536 $ARGS -- $FILES 2>/dev/null | tail -1)
  echo "R=$R $(echo $out | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["frames_per_s"], d["x_realtime_per_receiver"])')"
done; done
rm -rf $D
