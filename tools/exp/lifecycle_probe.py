# development: which FIB groups / CIFs the decoders pick up from the batched decodes under a few decoder life-cycle scripts (tests/cpp/mirror_lifecycle_driver)
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, oracle as O, stream_model as SM
O.build()
SUBS = {0: (0, 48, 2, 0), 1: (120, 27, 0, 1), 2: (200, 60, 2, 0), 3: (300, 24, 1, 0), 4: (400, 42, 1, 1), 5: (48, 72, 2, 0)}
subs = [O.subchannel(v[0], v[1], eep_level=v[2], eep_type=v[3]) for v in SUBS.values()]
iq, _ = SM.make_ensemble_stream(O, 22, subs, seed=77)
d = tempfile.mkdtemp()
iq.tofile(os.path.join(d, "iq.c32"))
env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "dab-radio_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
add = lambda f, i: f"{f} add {i} {SUBS[i][0]} {SUBS[i][1]} {SUBS[i][2]} {SUBS[i][3]}"
scripts = {"fic_from_0": ["0 fic 1"], "fic_from_15": ["15 fic 1"], "fic_9_off_11_on_15": ["9 fic 1", "11 fic 0", "15 fic 1"],
           "sub0_from_0": [add(0, 0)], "sub0_from_3_fic_from_0": ["0 fic 1", add(3, 0)], "sub_1_then_del_then_fic15": [add(1, 0), "4 del 0", "15 fic 1"],
           "five_at_15": [add(15, k) for k in range(5)] + ["15 fic 1"]}
for depth in ("3", "1"):
    for name, lines in scripts.items():
        open(os.path.join(d, "s.txt"), "w").write("\n".join(lines) + "\n")
        out = tempfile.mkdtemp()
        r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "mirror_lifecycle_driver"), os.path.join(d, "iq.c32"), out, "65536", os.path.join(d, "s.txt")],
                           capture_output=True, text=True, env=dict(env, DABGPU_MIRROR_DEPTH=depth))
        print(depth, name, r.stdout.strip(), r.stderr.strip()[-200:])
