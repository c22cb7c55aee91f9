#!/bin/bash
# SQ / GRBM / TCC counter passes of the channel decoder and the demodulator at 4096 ensembles (rocprofv3 --pmc, the program
# directly after `--`, counters in their own runs), reduced by tools/collect_counters.py to gpurun_out/counters_<tag>.json.
#   gpurun --timeout 1200 -- 'bash tools/prof_counters.sh v1'
set -u
TAG=${1:-vX}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
CMD="tools/bench_decode.py --ensembles 4096 --steps 2 --no-overlap --spb 75"     # (a fixed run length: every demodulator launch alike)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cnt_trace -o t -- python3 $CMD > $OUT/cnt_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/cnt_sq -o pmc -- python3 $CMD > $OUT/cnt_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cnt_sq2 -o pmc -- python3 $CMD > $OUT/cnt_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cnt_fetch -o pmc -- python3 $CMD > $OUT/cnt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cnt_write -o pmc -- python3 $CMD > $OUT/cnt_write.log 2>&1
python3 tools/collect_counters.py $OUT $TAG
rm -rf $OUT/cnt_trace $OUT/cnt_sq $OUT/cnt_sq2 $OUT/cnt_fetch $OUT/cnt_write
