export TMPDIR=/tmp
D=/tmp/abm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers 1 --frames 1200 --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_HARNESS_BENCH=1 LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib
for rep in 1 2; do for depth in 1 2 3 4 6; do
  mkdir -p $D/out; r=$(DABGPU_MIRROR_DEPTH=$depth taskset -c 64-127 ./tests/cpp/mirror_harness $D/rx0.c32 $D/out 65536 $ARGS 2>/dev/null | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["frames_per_s"])')
  echo "depth $depth $r"
done; done
rm -rf $D
