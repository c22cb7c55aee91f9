#!/usr/bin/env python3
"""Development bench: the frame session of the C ABI alone (dabgpu_frame_session_push_frame + fetches, 18 x 48 CU sub-channels) -- frames/s
when pushes are queued back to back, the host time of one push, and the latency of one frame pushed and fetched completely."""
import os
import sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dab-radio_amd"))
import numpy as np, dabgpu
rng = np.random.default_rng(1)
subs = [dabgpu.SubChannel(48 * k, 48, False, 0, 2, 0) for k in range(18)]
fs = dabgpu.FrameSession(0); fs.set_subchannels(subs)
frames = rng.integers(-127, 128, (8, 230400), dtype=np.int8)
for k in range(16): g = fs.push_frame(frames[k % 8])
fs.fetch_fib_group(g, 0)
t0 = time.perf_counter(); push = []
for k in range(200):
    a = time.perf_counter(); g = fs.push_frame(frames[k % 8]); push.append(time.perf_counter() - a)
fs.fetch_fib_group(g, 3); fs.fetch_cif(g, subs[17], 3)
dt = time.perf_counter() - t0
print("frames/s", 200 / dt, "x realtime", 200 / dt / 10.4167, "push median us", np.median(push) * 1e6, "p90", np.percentile(push, 90) * 1e6)
# latency of one frame: push + fetch everything
lat = []
for k in range(50):
    a = time.perf_counter(); g = fs.push_frame(frames[k % 8])
    for q in range(4): fs.fetch_fib_group(g, q)
    for c in range(4):
        for s in subs: fs.fetch_cif(g, s, c)
    lat.append(time.perf_counter() - a)
print("push + fetch all: median us", np.median(lat) * 1e6)
fs.close()
