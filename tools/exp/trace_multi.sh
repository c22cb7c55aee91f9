# experiment: merged kernel + memory-copy timeline of R receivers in one process, a 2 ms window in the steady state
R=${1:-4}; F=${2:-300}
export TMPDIR=/tmp
D=/tmp/tmm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers $R --frames $F --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
FILES=""; for k in $(seq 0 $((R-1))); do FILES="$FILES $D/rx$k.c32"; done
export DABGPU_DRIVER_BENCH=1 LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D/prof -o t -- ./tests/cpp/mirror_threads_driver 65536 $ARGS -- $FILES > $D/stdout.log 2>&1
tail -1 $D/stdout.log
kf=$(find $D/prof -name '*kernel_trace.csv' | head -1); mf=$(find $D/prof -name '*memory_copy_trace.csv' | head -1)
head -1 $mf
python3 - "$kf" "$mf" <<'PY'
import csv, sys
ks = [r for r in csv.DictReader(open(sys.argv[1]))]
ms = [r for r in csv.DictReader(open(sys.argv[2]))]
def short(n):
    n = n.split('(')[0].split('<')[0]; return n[n.rfind('::') + 2:] if '::' in n else n
ev = []
for r in ks:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), f"q{r['Queue_Id']} {short(r['Kernel_Name']) or 'blit'}"))
for r in ms:
    size = r.get('Bytes') or r.get('Size') or '?'
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), f"   copy {r.get('Direction', '?')} {size} B"))
ev.sort()
mid = ev[len(ev) // 2][0]
base = mid
for s, e, what in ev:
    if mid <= s < mid + 2_000_000:
        print(f"{(s - base) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {what}")
PY
rm -rf $D
