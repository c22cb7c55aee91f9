#!/bin/bash
# The headline workload's evidence only (bench line, rocprofv3 kernel stats, the two HBM counter passes), reduced like tools/profile_round.sh does:
#   gpurun --timeout 1200 -- 'bash tools/profile_headline.sh v2 r06'    then    bash tools/publish_profiles.sh v2 r06
set -u
TAG=${1:-vX}
ROUND=${2:-r06}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-cpu-baseline --no-check --no-extras"
python3 bench.py > $OUT/bench_n1_$TAG.json 2> $OUT/bench_n1_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 20 --warmup 3 $BENCH_ARGS > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_write.log 2>&1
python3 tools/collect_profiles.py $OUT $TAG $ROUND > $OUT/collect_$TAG.log 2>&1
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
python3 tools/rss_floor.py > $OUT/rss_floor_$TAG.json 2> $OUT/rss_floor_$TAG.err
tail -3 $OUT/collect_$TAG.log
