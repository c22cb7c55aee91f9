#!/usr/bin/env python3
"""Soak of the batch interfaces: `--steps` steps of bench.py's Pipeline (sync + demod + FIC + MSC of E ensembles, two frames in flight); the process's
resident set and the device's used memory are sampled at 25 % / 60 % / 100 %: growth = a leak (tools/soak_mirror.py does the same for the drop-in classes).

    python tools/soak_batch.py [--ensembles 256] [--steps 4000]
"""
import argparse, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")]
import torch, dabgpu, bench

ap = argparse.ArgumentParser()
ap.add_argument("--ensembles", type=int, default=256)
ap.add_argument("--steps", type=int, default=4000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = dabgpu.Context(0)
p = bench.Pipeline(ctx, dabgpu, torch, dev, a.ensembles, min(a.ensembles, 16), seed=7, inflight=2, layout=1, synced=True)
p.fill()


def mem():
    rss = int(re.search(r"VmRSS:\s+(\d+)", open("/proc/self/status").read()).group(1)) / 1024
    free, total = torch.cuda.mem_get_info()
    return {"host_rss_MB": round(rss, 1), "device_used_MB": round((total - free) / 2**20, 1)}


marks = {}
for k in range(a.steps):
    p.step()
    if k in (a.steps // 4, a.steps * 6 // 10, a.steps - 1):
        torch.cuda.synchronize()
        marks[k] = mem()
chk = p.check(dabgpu)
m = [marks[k] for k in sorted(marks)]
out = {"what": "bench.Pipeline.step (sync + demod + FIC + MSC, two frames in flight)", "ensembles": a.ensembles, "steps": a.steps, "frames": a.ensembles * a.steps,
       "memory_at_25_percent": m[0], "memory_at_60_percent": m[1], "memory_at_end": m[2], "check": {k: v for k, v in chk.items() if not isinstance(v, (list, dict))}}
out["ok"] = bool(m[2]["host_rss_MB"] <= m[0]["host_rss_MB"] + 8 and m[2]["device_used_MB"] <= m[0]["device_used_MB"] + 8)
print(json.dumps(out))
sys.exit(0 if out["ok"] else 1)
