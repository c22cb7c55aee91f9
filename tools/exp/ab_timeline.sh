# development: kernel trace of one receiver (mirror_threads_driver, timing mode) with build/exp/old/libdabgpu.so and with the tree's, same box
export TMPDIR=/tmp
D=/tmp/abt; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers 1 --frames 600 --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_DRIVER_BENCH=1
for which in old new old new; do
  if [ $which = old ]; then export LD_LIBRARY_PATH=build/exp/old:/opt/rocm/lib; else export LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib; fi
  rm -rf $D/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o t -- ./tests/cpp/mirror_threads_driver 65536 $ARGS -- $D/rx0.c32 > $D/stdout.log 2>&1
  echo "== $which $(tail -1 $D/stdout.log | cut -c1-120)"
  f=$(find $D/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dabgpu' in r['Name'] and int(r['Calls']) > 100:
        print("   %-50s %5s x %8.1f us" % (r['Name'].split('(')[0].replace('void ', '').replace('dabgpu::', '')[:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
rm -rf $D
