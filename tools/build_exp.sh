#!/bin/bash
# development: build/exp/libdabgpu_<tag>.so = the product objects with ONE translation unit rebuilt with extra flags
#   tools/build_exp.sh <source.hip> <tag>=<flags> [<tag>=<flags> ...]      e.g.  tools/build_exp.sh ofdm_demod.hip e4=-DDABGPU_EXP=4
# The product sources carry no ablation switches.  The timing-only / instrumented variants (-DDABGPU_EXP=<bits>, -DDABGPU_PRIO=<hex>)
# live in tools/exp/<name>_exp.hip: snapshots of the round-3 kernels with their switches (what profiles/r02 and r03 ab_notes.md were
# measured with; tools/kphase.py, tools/abl_*.sh).  A <flags> that mentions DABGPU_EXP or DABGPU_PRIO is built from the snapshot.
set -u
cd "$(dirname "$0")/../dab-radio_amd/csrc"
SRC=$1; shift
OUT=../../build/exp; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -I. -Wall -Wno-unused-function"
OTHERS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  IN=$SRC
  case "$flags" in *DABGPU_EXP*|*DABGPU_PRIO*) [ -f ../../tools/exp/${SRC%.hip}_exp.hip ] && IN=../../tools/exp/${SRC%.hip}_exp.hip;; esac
  ( hipcc $F $flags -c $IN -o $OUT/${SRC%.hip}_$tag.o 2>$OUT/$tag.log && hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdabgpu_$tag.so $OUT/${SRC%.hip}_$tag.o $OTHERS 2>>$OUT/$tag.log || { echo "FAILED $tag"; grep -A6 "error" $OUT/$tag.log | head -20; } ) &
  while (( $(jobs -r | wc -l) >= 7 )); do wait -n; done
done
wait
ls $OUT/*.so | wc -l
