# development: the receiver bank at several depths of rounds / gathering windows (tools/bench_mirror_multi.py), REPS runs each, sorted frames/s per setting
for rounds in ${ROUNDS:-1 2 3}; do for g in ${GATHER:-0 1000}; do
  for rx in ${RX:-8 32}; do
    vals=""
    for rep in $(seq ${REPS:-3}); do
      v=$(DABGPU_BANK_ROUNDS=$rounds DABGPU_BANK_GATHER_US=$g DABGPU_MIRROR_BANK=1 python tools/bench_mirror_multi.py --receivers $rx 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(int(d['runs'][0]['frames_per_s']))")
      vals="$vals $v"
    done
    echo "rounds $rounds gather $g rx $rx: $(echo $vals | tr ' ' '\n' | sort -n | tr '\n' ' ')"
  done
done; done
