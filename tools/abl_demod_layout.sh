for t in base e32 e64 e96; do
  for spb in 25 75; do DABGPU_LIB=$PWD/build/exp/libdabgpu_$t.so python tools/bench_demod_layout.py --spb $spb 2>/dev/null | tail -1; done
done
