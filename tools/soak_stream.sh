#!/bin/bash
# development: N runs of tools/bench_stream.py --digest with the given arguments; counts how often each (input checksum, streams that lost
# synchronisation, frames, frames of the first call) occurs -- one line = deterministic
N=${1:-30}; shift
declare -A seen
for i in $(seq 1 $N); do
  r=$(python tools/bench_stream.py --digest "$@" 2>/dev/null | tail -2 | python3 -c "
import sys,json
a=json.loads(sys.stdin.readline()); d=json.loads(sys.stdin.readline())
print(a['input_digest'], a['streams_with_desync'], d['frames_total'], d['per_call_frames'][0])")
  seen["$r"]=$(( ${seen["$r"]:-0} + 1 ))
done
for k in "${!seen[@]}"; do echo "$k : ${seen[$k]}"; done
