# development: A/B builds of ofdm_demod.hip (tools/build_exp.sh ofdm_demod.hip base=-DDABGPU_EXP=0 e128=-DDABGPU_EXP=128 ...) through
# tools/bench_demod_layout.py on synthetic OFDM frames
for t in ${TAGS:-base e128}; do
  for spb in 38 75; do DABGPU_LIB=$PWD/build/exp/libdabgpu_$t.so python tools/bench_demod_layout.py --data ofdm --spb $spb 2>/dev/null | tail -1; done
done
