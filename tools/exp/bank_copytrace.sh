# development: rocprofv3 kernel + memory-copy trace of N receivers of one process (tests/cpp/mirror_threads_driver in its timing mode)
#   gpurun -- 'RX=8 BANK=1 bash tools/exp/bank_copytrace.sh'
set -u
RX=${RX:-8}; BANK=${BANK:-1}
export TMPDIR=/tmp
D=/tmp/bank_ct; mkdir -p $D gpurun_out/r06
python3 tools/bench_mirror_multi.py --receivers $RX --frames 150 --only-write $D > /dev/null 2>&1
ARGS="65536"; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
FILES=""; for k in $(seq 0 $((RX-1))); do FILES="$FILES $D/rx$k.c32"; done
export LD_LIBRARY_PATH=$PWD/dab-radio_amd:/opt/rocm/lib:${LD_LIBRARY_PATH:-}
export DABGPU_DRIVER_BENCH=1 DABGPU_MIRROR_BANK=$BANK
rm -rf $D/prof
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D/prof -o t -- ./tests/cpp/mirror_threads_driver $ARGS -- $FILES > $D/run.log 2>&1
tail -1 $D/run.log | cut -c1-200
python3 - <<PY
import csv, glob, collections
f = glob.glob("$D/prof/**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    by = collections.defaultdict(list)
    for r in rows:
        n = int(r.get("Bytes", r.get("Size", 0)) or 0) if ("Bytes" in r or "Size" in r) else 0
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        by[(r["Direction"], n)].append(d)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
        v.sort()
        print(k, "n", len(v), "median us", round(v[len(v)//2], 1), "p90", round(v[int(len(v)*0.9)], 1), "GB/s at median", round(k[1] / max(v[len(v)//2], 1e-9) / 1e3, 1) if k[1] else "")
    print("columns:", list(rows[0].keys()))
f = glob.glob("$D/prof/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
