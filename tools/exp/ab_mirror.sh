# development: A/B of two builds behind one capture on one box (interleaved runs): build/exp/old/{mirror_harness, libdabgpu.so (optional)} against the tree's
export TMPDIR=/tmp
D=/tmp/abm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers 1 --frames 1200 --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_HARNESS_BENCH=1 DABGPU_MIRROR_PROFILE=1
for rep in 1 2 3 4; do
  for which in old new; do
    H=./tests/cpp/mirror_harness; export LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib
    if [ $which = old ]; then
      [ -x build/exp/old/mirror_harness ] && H=build/exp/old/mirror_harness
      [ -f build/exp/old/libdabgpu.so ] && export LD_LIBRARY_PATH=build/exp/old:/opt/rocm/lib
    fi
    mkdir -p $D/out; r=$(taskset -c 64-127 $H $D/rx0.c32 $D/out 65536 $ARGS 2>$D/err.txt | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["frames_per_s"])')
    echo "$which $r $(grep -o "reader {[^}]*}" $D/err.txt)"
  done
done
rm -rf $D
