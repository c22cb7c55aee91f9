for r in 1 2; do for t in base e256 e768; do
  for F in 1024 4096; do DABGPU_LIB=$PWD/build/exp/libdabgpu_$t.so python tools/bench_demod_layout.py --data ofdm --spb 75 --frames $F --reps 60 2>/dev/null | tail -1; done
done; done
