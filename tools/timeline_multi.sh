#!/bin/bash
# development: do the kernels of several receivers in one process (tests/cpp/mirror_threads_driver, timing mode) overlap on the device?
# kernel trace of R receivers x F frames -> hardware queues used, how many trellis launches run at once, busy time per queue
#   gpurun -- 'bash tools/timeline_multi.sh 4 300 > gpurun_out/timeline_multi.txt'
R=${1:-4}; F=${2:-300}
export TMPDIR=/tmp
D=gpurun_out/tmm; rm -rf $D; mkdir -p $D
python3 tools/bench_mirror_multi.py --receivers $R --frames $F --only-write $D || exit 1
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
FILES=""; for k in $(seq 0 $((R-1))); do FILES="$FILES $D/rx$k.c32"; done
export DABGPU_DRIVER_BENCH=1 LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --output-format csv -d $D/prof -o t -- ./tests/cpp/mirror_threads_driver 65536 $ARGS -- $FILES > $D/stdout.log 2>&1
tail -1 $D/stdout.log
f=$(find $D/prof -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'dabgpu' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# steady state: the middle half
t0, t1 = rows[len(rows) // 4]['s'], rows[3 * len(rows) // 4]['s']
mid = [r for r in rows if t0 <= r['s'] < t1]
span = (t1 - t0) / 1e3
print(f"{len(rows)} launches; steady-state window {span / 1e3:.1f} ms, {len(mid)} launches")
q = collections.defaultdict(lambda: [0, 0.0])
for r in mid:
    q[r['Queue_Id']][0] += 1; q[r['Queue_Id']][1] += (r['e'] - r['s']) / 1e3
print("hardware queues:", len(q))
for k, (n, busy) in sorted(q.items()):
    print(f"  queue {k}: {n} launches, busy {busy / span * 100:.1f} % of the window")
def short(n):
    n = n.split('(')[0]; return n[n.rfind('::') + 2:] if '::' in n else n
kn = collections.defaultdict(lambda: [0, 0.0])
for r in mid:
    kn[short(r['Kernel_Name'])][0] += 1; kn[short(r['Kernel_Name'])][1] += (r['e'] - r['s']) / 1e3
for k, (n, us) in sorted(kn.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:60]:60s} {n:6d} x {us / n:8.1f} us = {us / span * 100:5.1f} % of the window")
# concurrency: how many launches are in flight, time-weighted
ev = sorted([(r['s'], 1) for r in mid] + [(r['e'], -1) for r in mid])
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print("launches in flight (share of the window):", {k: round(v / tot * 100, 1) for k, v in sorted(hist.items())})
big = [r for r in mid if 'vit' in r['Kernel_Name']]
ev = sorted([(r['s'], 1) for r in big] + [(r['e'], -1) for r in big])
lvl, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print("trellis launches in flight:", {k: round(v / tot * 100, 1) for k, v in sorted(hist.items())})
# do the trellis launches slow down when they overlap?
import statistics
print("trellis launch duration: median %.1f us, p90 %.1f us" % (statistics.median((r['e'] - r['s']) / 1e3 for r in big), sorted((r['e'] - r['s']) / 1e3 for r in big)[int(len(big) * .9)]))
PY
rm -rf $D
