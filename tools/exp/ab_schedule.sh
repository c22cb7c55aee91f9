for N in 256 512 768 1024 2048; do for S in rounds resident; do for F in c32 raw_u8; do
python tools/bench_stream.py --streams $N --block-frames 4 --retained --calls 7 --format $F --schedule $S 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$N', '$S', '$F', round(d['steady_frames_per_s']), d['per_call_ms'][2:])"
done; done; done
