#!/bin/bash
# Round profile pass on the GPU box: the bench line, rocprofv3 kernel stats + the two HBM counter passes of the headline workload,
# kernel stats of the sync-enabled configs[3]/[4] step and of the device-resident chain at 4096 ensembles, and the side benches.
# Summaries land under gpurun_out/ (scratch); tools/publish_profiles.sh copies what profiles/<round>/ keeps.
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh v1 r04'
set -u
TAG=${1:-vX}
ROUND=${2:-r04}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-cpu-baseline --no-check --no-extras"
python3 bench.py > $OUT/bench_n1_$TAG.json 2> $OUT/bench_n1_$TAG.err
# (symbols_per_block = 0 resolves to what dabgpu_ofdm_tune records in each process's warm-up)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 20 --warmup 3 $BENCH_ARGS > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_write.log 2>&1
# the path SURVEY 8(d) defines for configs 3 / 4: PRS sync + demod at the tracked offsets + FIC + MSC, 4096 ensembles, two frames in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sync -o trace -- python3 bench.py --workload full --ensembles 4096 --steps 8 --warmup 2 --no-cpu-baseline --no-check > $OUT/prof_sync.log 2>&1
cp $(find $OUT/prof_sync -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_synced4096_$TAG.csv; rm -rf $OUT/prof_sync
# the device-resident chain: unsynchronised raw_u8 streams -> stream bank -> rings -> FIC + MSC -> DAB+, 4096 ensembles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_chain -o trace -- python3 tools/bench_chain.py --ensembles 4096 --steps 8 > $OUT/bench_chain_profiled_$TAG.json 2> $OUT/prof_chain.log
cp $(find $OUT/prof_chain -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_chain4096_$TAG.csv; rm -rf $OUT/prof_chain
python3 tools/bench_chain.py --ensembles 4096 > $OUT/bench_chain_$TAG.json 2> $OUT/bench_chain_$TAG.err
python3 bench.py --workload full --no-cpu-baseline > $OUT/bench_full_$TAG.json 2> $OUT/bench_full_$TAG.err
python3 tools/bench_decode.py --ensembles 4096 --steps 6 > $OUT/bench_decode_4096_$TAG.json 2> $OUT/bench_decode_4096_$TAG.err
python3 tools/bench_mirror.py > $OUT/bench_mirror_$TAG.json 2> $OUT/bench_mirror_$TAG.err
# several receivers in one process: private pipelines, all in the receiver bank, and the classes' AUTO rule (the second and later ones banked)
for B in 0 1; do DABGPU_BANK_PROFILE=1 DABGPU_MIRROR_BANK=$B python3 tools/bench_mirror_multi.py --receivers 1 2 4 8 16 32 > $OUT/bench_mirror_multi_bank${B}_$TAG.json 2> $OUT/bench_mirror_multi_bank${B}_$TAG.err; done
DABGPU_BANK_PROFILE=1 python3 tools/bench_mirror_multi.py --receivers 2 4 8 16 32 > $OUT/bench_mirror_multi_auto_$TAG.json 2> $OUT/bench_mirror_multi_auto_$TAG.err
python3 tools/soak_mirror.py --frames 300 --repeats 40 > $OUT/soak_mirror_$TAG.json 2> $OUT/soak_mirror_$TAG.err
python3 tools/bench_stream.py --streams 1024 --block-frames 4 > $OUT/bench_stream_1024x4_$TAG.json 2> $OUT/bench_stream_1024x4_$TAG.err
{ python3 tools/bench_stream.py --streams 1024 --block-frames 4 --retained; python3 tools/bench_stream.py --streams 1024 --block-frames 4 --format raw_u8 --retained; python3 tools/bench_stream.py --retained; } > $OUT/bench_stream_retained_$TAG.json 2> $OUT/bench_stream_retained_$TAG.err
for F in 1024 4096 16384; do python3 tools/bench_fic.py --frames $F; done > $OUT/bench_fic_$TAG.json 2> $OUT/bench_fic_$TAG.err
python3 tools/bench_sync.py > $OUT/bench_sync_$TAG.json 2> $OUT/bench_sync_$TAG.err
LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib ./tests/cpp/multi_gpu_harness --devices 0 --ensembles 8192 --steps 10 > $OUT/bench_cpp_host_$TAG.json 2> $OUT/bench_cpp_host_$TAG.err
LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib ./tests/cpp/multi_gpu_harness --devices 0,0,0,0,0,0,0,0 --ensembles 1024 --steps 20 --distinct 16 >> $OUT/bench_cpp_host_$TAG.json 2>> $OUT/bench_cpp_host_$TAG.err
LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib ./tests/cpp/multi_gpu_harness --devices 0 --ensembles 8192 --steps 10 --aligned >> $OUT/bench_cpp_host_$TAG.json 2>> $OUT/bench_cpp_host_$TAG.err
python3 tools/bench_io.py > $OUT/bench_io_$TAG.json 2> $OUT/bench_io_$TAG.err
python3 tools/bench_dabplus.py > $OUT/bench_dabplus_$TAG.json 2> $OUT/bench_dabplus_$TAG.err
python3 tools/bench_ingest.py > $OUT/bench_ingest_$TAG.json 2> $OUT/bench_ingest_$TAG.err
# SQ / GRBM / TCC counters of the decoder and the demodulator at 4096 ensembles (own passes, program directly after --)
bash tools/prof_counters.sh $TAG > $OUT/prof_counters_$TAG.log 2>&1
# keep the merge-back small: reduce on the box, then drop the raw dumps
python3 tools/collect_profiles.py $OUT $TAG $ROUND > $OUT/collect_$TAG.log 2>&1
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write
du -sh $OUT
