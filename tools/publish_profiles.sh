#!/bin/bash
# After `gpurun -- 'bash tools/profile_round.sh <tag> <round>'` has merged its summaries into gpurun_out/ (scratch): copy what
# profiles/<round>/ keeps (tracked) and refresh profiles/hbm_traffic.json, which bench.py reports as roofline.traffic.
#   bash tools/publish_profiles.sh v2 r03
set -eu
TAG=$1; ROUND=$2
cd "$(dirname "$0")/.."
D=profiles/$ROUND; S=gpurun_out/summary_$TAG
mkdir -p $D
cp $S/kernel_stats_$TAG.csv $S/pmc_fetch_size_$TAG.csv $S/pmc_write_size_$TAG.csv $D/
[ -f $S/hbm_traffic_decode_$TAG.json ] && cp $S/hbm_traffic_decode_$TAG.json $D/
cp $S/hbm_traffic.json $D/hbm_traffic_$TAG.json
cp $S/hbm_traffic.json profiles/hbm_traffic.json
for f in counters kernel_stats_decode4096 kernel_stats_synced4096 kernel_stats_chain4096; do for e in json csv; do [ -f gpurun_out/${f}_$TAG.$e ] && cp gpurun_out/${f}_$TAG.$e $D/; done; done
for f in bench_n1 bench_decode bench_decode_4096 bench_decode_4096_natural bench_full bench_ingest bench_mirror bench_stream bench_stream_1024x4 bench_stream_retained bench_io bench_fic \
         bench_cpp_host bench_dabplus bench_chain bench_chain_profiled bench_sync bench_mirror_multi_bank0 bench_mirror_multi_bank1 bench_mirror_multi_auto soak_mirror; do
  [ -s gpurun_out/${f}_$TAG.json ] && cp gpurun_out/${f}_$TAG.json $D/
done
ls $D | grep "_$TAG" | wc -l
