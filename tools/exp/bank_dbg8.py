import os, sys, subprocess, json, numpy as np
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
mode = sys.argv[1] if len(sys.argv)>1 else "1"
extra = dict(a.split("=") for a in sys.argv[2:])
bad=0
for trial in range(12):
    d='/tmp/bk8/dump%d'%trial; os.makedirs(d, exist_ok=True)
    res=subprocess.run(args+['--']+paths, capture_output=True, text=True, env=dict(env, DABGPU_MIRROR_BANK=mode, DABGPU_DRIVER_DUMP=d, **extra), timeout=600)
    out=json.loads(res.stdout.strip().splitlines()[-1])
    flags=[r["threaded_equals_serial"] for r in out["per_receiver"]]
    if not all(flags):
        bad+=1
        for r,f in enumerate(flags):
            if f: continue
            for kind,rec in (("fibs",30),("msc",1)):
                a=np.fromfile(f'{d}/serial_{kind}_{r}.bin',np.uint8); b=np.fromfile(f'{d}/threaded_{kind}_{r}.bin',np.uint8)
                n=min(a.size,b.size); diff=np.nonzero(a[:n]!=b[:n])[0]
                print("trial",trial,"rx",r,kind,"sizes",a.size,b.size,"first diffs",diff[:8], "n diff", diff.size, "last", diff[-3:] if diff.size else None)
print("mode",mode,extra,"bad trials",bad,"of 12")
