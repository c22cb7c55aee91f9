#!/bin/bash
# Round profile pass on the GPU box: kernel stats + the two HBM counter passes of the bench workload, and the
# side benches.  Summaries land under gpurun_out/ (scratch); tools/collect_profiles.py copies them to profiles/.
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh v1 r03'
set -u
TAG=${1:-vX}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-cpu-baseline --no-check --no-extras"
ROUND=${2:-r03}
python3 bench.py > $OUT/bench_n1_$TAG.json 2> $OUT/bench_n1_$TAG.err
# the profiled runs use the workgroup size the bench run chose on this box (its set-up timing would add launches of the other sizes)
# (symbols_per_block = 0: the library's own one-time calibration picks the run length in every process)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 20 --warmup 3 $BENCH_ARGS > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 $BENCH_ARGS > $OUT/pmc_write.log 2>&1
python3 tools/bench_decode.py > $OUT/bench_decode_$TAG.json 2> $OUT/bench_decode_$TAG.err
python3 tools/bench_decode.py --ensembles 4096 --steps 6 > $OUT/bench_decode_4096_$TAG.json 2> $OUT/bench_decode_4096_$TAG.err
python3 bench.py --workload full --no-cpu-baseline > $OUT/bench_full_$TAG.json 2> $OUT/bench_full_$TAG.err
python3 tools/bench_ingest.py > $OUT/bench_ingest_$TAG.json 2> $OUT/bench_ingest_$TAG.err
python3 tools/bench_mirror.py > $OUT/bench_mirror_$TAG.json 2> $OUT/bench_mirror_$TAG.err
python3 tools/bench_stream.py --streams 1024 --block-frames 4 > $OUT/bench_stream_1024x4_$TAG.json 2> $OUT/bench_stream_1024x4_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dec -o trace -- python3 tools/bench_decode.py --ensembles 4096 --steps 4 > $OUT/prof_dec.log 2>&1
cp $(find $OUT/prof_dec -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_decode4096_$TAG.csv; rm -rf $OUT/prof_dec
for L in classed natural; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_dec_fetch_$L -o pmc -- python3 tools/bench_decode.py --ensembles 1024 --steps 2 --hist-layout $L > $OUT/pmc_dec.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_dec_write_$L -o pmc -- python3 tools/bench_decode.py --ensembles 1024 --steps 2 --hist-layout $L > $OUT/pmc_dec.log 2>&1
done
python3 tools/bench_decode.py --ensembles 4096 --steps 6 --hist-layout natural > $OUT/bench_decode_4096_natural_$TAG.json 2> $OUT/bench_decode_4096_natural_$TAG.err
python3 tools/bench_io.py > $OUT/bench_io_$TAG.json 2> $OUT/bench_io_$TAG.err
for F in 1024 4096 16384; do python3 tools/bench_fic.py --frames $F; done > $OUT/bench_fic_$TAG.json 2> $OUT/bench_fic_$TAG.err
LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib ./tests/cpp/multi_gpu_harness --devices 0 --ensembles 8192 --steps 10 > $OUT/bench_cpp_host_$TAG.json 2> $OUT/bench_cpp_host_$TAG.err
LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib ./tests/cpp/multi_gpu_harness --devices 0,0 --ensembles 4096 --steps 10 >> $OUT/bench_cpp_host_$TAG.json 2>> $OUT/bench_cpp_host_$TAG.err
# SQ / GRBM / TCC counters of the decoder and the demodulator at 4096 ensembles (own passes, program directly after --)
bash tools/prof_counters.sh $TAG > $OUT/prof_counters_$TAG.log 2>&1
python3 tools/bench_stream.py > $OUT/bench_stream_$TAG.json 2> $OUT/bench_stream_$TAG.err
# retained blocks (no carry-over copy): c32 and raw_u8 at 1024 x 4 frames, c32 at 256 x 2
{ python3 tools/bench_stream.py --streams 1024 --block-frames 4 --retained; python3 tools/bench_stream.py --streams 1024 --block-frames 4 --format raw_u8 --retained; \
  python3 tools/bench_stream.py --streams 1024 --block-frames 4 --format raw_u8; python3 tools/bench_stream.py --retained; } > $OUT/bench_stream_retained_$TAG.json 2> $OUT/bench_stream_retained_$TAG.err
python3 tools/bench_dabplus.py > $OUT/bench_dabplus_$TAG.json 2> $OUT/bench_dabplus_$TAG.err
# keep the merge-back small: reduce on the box, then drop the raw dumps
python3 tools/collect_profiles.py $OUT $TAG $ROUND > $OUT/collect_$TAG.log 2>&1
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_dec_*
du -sh $OUT
