# development: the bank's per-round profile lines and one member's reader / delivery profile at a given depth of rounds
for rounds in ${ROUNDS:-1 2}; do echo "== rounds $rounds"; DABGPU_DRIVER_CPU=1 DABGPU_MIRROR_PROFILE=1 DABGPU_BANK_ROUNDS=$rounds DABGPU_BANK_GATHER_US=${GATHER:-1000} DABGPU_BANK_PROFILE=1 DABGPU_MIRROR_BANK=1 python tools/bench_mirror_multi.py --receivers ${RX:-32} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']:
    print(r['receivers'], r['frames_per_s'], r.get('host_cpu_ms_per_frame'))
    ps=r.get('profile',[])
    for p in ps:
        if not p.startswith('OFDM_Demod'): print('   ',p)
    for p in [q for q in ps if q.startswith('OFDM_Demod')][:3]: print('   ',p)
"; done
