import os, sys, subprocess, json
ROOT='/root/repo'
sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/oracle')
import oracle as O, stream_model as SM
O.build()
subs = [O.subchannel(0, 24, eep_level=2, eep_type=0), O.subchannel(60, 21, eep_level=1, eep_type=1)]
paths=[]
os.makedirs('/tmp/bk', exist_ok=True)
for k in range(2):
    stream,_ = SM.make_ensemble_stream(O, 9, subs, seed=900+k, cfo=(1.1e-3,-2.4e-3)[k], timing_pad=(300,4321)[k], noise=2.0)
    p='/tmp/bk/rx%d.c32'%k; stream.tofile(p); paths.append(p)
args=[ROOT+'/tests/cpp/mirror_threads_driver','65536']
for s in subs: args += [str(s.start_address), str(s.length), str(s.eep_prot_level), str(s.eep_type)]
env=dict(os.environ); env['LD_LIBRARY_PATH']=ROOT+'/dab-radio_amd:/opt/rocm/lib:'+env.get('LD_LIBRARY_PATH','')
for trial in range(3):
    res=subprocess.run(args+['--']+paths, capture_output=True, text=True, env=env, timeout=600)
    print(res.returncode, res.stdout, res.stderr[-500:])
