# development: the receiver bank at several depths of rounds / gathering windows (tools/bench_mirror_multi.py)
for rounds in ${ROUNDS:-1 2 3}; do for g in ${GATHER:-0 1000}; do echo "== rounds $rounds gather $g"; DABGPU_BANK_ROUNDS=$rounds DABGPU_BANK_GATHER_US=$g DABGPU_BANK_PROFILE=1 DABGPU_MIRROR_BANK=1 python tools/bench_mirror_multi.py --receivers ${RX:-8 16 32} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']:
    print(r['receivers'], r['frames_per_s'], r.get('host_cpu_ms_per_frame'), [p[p.find('rounds,')-6:] for p in r.get('profile',[]) if p.startswith('receiver bank')])
"; done; done
