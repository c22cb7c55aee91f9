import os, sys, subprocess, json
ROOT='/root/repo'
sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/oracle')
import oracle as O, stream_model as SM
O.build()
subs = [O.subchannel(0, 48, eep_level=2, eep_type=0), O.subchannel(100, 58, is_uep=True, uep_index=29), O.subchannel(300, 42, eep_level=1, eep_type=1)]
paths=[]
os.makedirs('/tmp/bk8', exist_ok=True)
for k in range(8):
    stream,_ = SM.make_ensemble_stream(O, 8, subs, seed=1200+k, cfo=(-2.2e-3+0.6e-3*k), timing_pad=137*k+11, noise=2.0)
    p='/tmp/bk8/rx%d.c32'%k; stream.tofile(p); paths.append(p)
args=[ROOT+'/tests/cpp/mirror_threads_driver','65536']
for s in subs:
    if s.is_uep: continue
    args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
env=dict(os.environ); env['LD_LIBRARY_PATH']=ROOT+'/dab-radio_amd:/opt/rocm/lib:'+env.get('LD_LIBRARY_PATH','')
for bank in ("0","1","1"):
    res=subprocess.run(args+['--']+paths, capture_output=True, text=True, env=dict(env, DABGPU_MIRROR_BANK=bank, DABGPU_BANK_PROFILE="1"), timeout=600)
    print(bank, res.returncode, res.stdout[-3000:], res.stderr[-1500:])
