#!/bin/bash
# Round profile pass on the GPU box: kernel stats + the two HBM counter passes of the bench workload, and the
# side benches.  Summaries land under gpurun_out/ (scratch); tools/collect_profiles.py copies them to profiles/.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh v5'
set -u
TAG=${1:-vX}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-cpu-baseline --no-check"
python3 bench.py > $OUT/bench_n1_$TAG.json 2> $OUT/bench_n1_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 20 --warmup 3 $BENCH_ARGS > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_write.log 2>&1
python3 tools/bench_decode.py > $OUT/bench_decode_$TAG.json 2> $OUT/bench_decode_$TAG.err
python3 tools/bench_io.py > $OUT/bench_io_$TAG.json 2> $OUT/bench_io_$TAG.err
python3 tools/bench_stream.py > $OUT/bench_stream_$TAG.json 2> $OUT/bench_stream_$TAG.err
python3 tools/bench_dabplus.py > $OUT/bench_dabplus_$TAG.json 2> $OUT/bench_dabplus_$TAG.err
# keep the merge-back small: reduce on the box, then drop the raw dumps
python3 tools/collect_profiles.py $OUT $TAG > $OUT/collect_$TAG.log 2>&1
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
du -sh $OUT
