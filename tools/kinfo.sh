#!/bin/bash
# development: VGPR / SGPR / scratch / LDS of the kernels of one source file under extra flags (code-object metadata of a -S compile)
#   tools/kinfo.sh ofdm_demod.hip "ofdm_demod_kernelILb0ELi0ELb0ELb0E" [flags...]
cd "$(dirname "$0")/../dab-radio_amd/csrc"
SRC=$1; PAT=$2; shift 2
OUT=/tmp/isa/kinfo_$$.s; mkdir -p /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -I. "$@" -S --cuda-device-only $SRC -o $OUT 2>/dev/null
awk -v pat="$PAT" '/^  - \.agpr_count|^  - \.args/{blk=""} {blk=blk"\n"$0} /\.name:/{name=$2} /\.vgpr_spill_count/{ if (name ~ pat) { print name; n=split(blk, L, "\n"); for(i=1;i<=n;i++) if (L[i] ~ /vgpr_count|sgpr_count|private_segment_fixed|vgpr_spill|group_segment_fixed/) print "   " L[i] } }' $OUT
[ -n "$KEEP" ] && cp $OUT $KEEP
rm -f $OUT
