# development: several receivers in one process, build/exp/old/{libdabgpu.so, mirror_threads_driver} against the tree's, one box, interleaved, pinned
export TMPDIR=/tmp
D=/tmp/abmm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers 8 --frames 1800 --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_DRIVER_BENCH=1
for rep in 1 2; do for R in 1 2 4 8; do for which in old new; do
  H=./tests/cpp/mirror_threads_driver; export LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib
  if [ $which = old ]; then H=build/exp/old/mirror_threads_driver; export LD_LIBRARY_PATH=build/exp/old:/opt/rocm/lib; fi
  FILES=""; for k in $(seq 0 $((R-1))); do FILES="$FILES $D/rx$k.c32"; done
  out=$(taskset -c 64-127 $H 65536 $ARGS -- $FILES 2>/dev/null | tail -1)
  echo "$which R=$R $(echo $out | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["frames_per_s"], d["x_realtime_per_receiver"])')"
done; done; done
rm -rf $D
