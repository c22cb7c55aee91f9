# development: timing ablations of vit_octet_kernel (tools/build_exp.sh viterbi_octet.hip <tag>=-DDABGPU_EXP=<bits> ...)
for t in "$@"; do
  for f in 4096 8192; do
    DABGPU_LIB=$PWD/build/exp/libdabgpu_$t.so python tools/bench_fic.py --frames $f --mappings 3 --no-check --reps 30 2>/dev/null | tail -1
  done
done
