#!/usr/bin/env python3
"""Soak of ONE receiver behind the drop-in classes: a raw_u8 capture (the reference app's default input format, app_iq_readers.h:23-30) of
`--frames` transmission frames preceded by a stretch of noise is piped `--repeats` times back to back into dab-radio_amd/host/apps/dabgpu_radio_cli
(--configuration dab+ofdm, 18 sub-channels) -- every repetition ends in a break of the framing, so the receiver loses lock, resets its
frequency state and re-acquires (ofdm_demodulator.cpp:277-289, :291-347) `--repeats` times while frames are in flight -- and the process's resident
set and the device's used memory are sampled while it runs.  Reports frames, FIB CRCs, sub-channel bytes per repetition (all repetitions must
deliver the same), frames/s over the whole run and the memory at 25 % and 100 % of the run (growth = a leak).

    python tools/soak_mirror.py [--frames 300] [--repeats 40]
"""
import argparse, json, os, re, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, dabgpu, dabsynth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--repeats", type=int, default=40)
ap.add_argument("--subchannels", type=int, default=18)
ap.add_argument("--cli", default=None, help="another build of dabgpu_radio_cli (A/B of builds)")
ap.add_argument("--libdir", default=None, help="directory of the libdabgpu.so that build runs on")
a = ap.parse_args()
dev = torch.device("cuda", 0)
prs, mapper, _ = dabgpu.host_tables()
mux = dabsynth.Multiplex(1, 21, dev)
f2 = dabsynth.modulate(mux.frame_bits[0], prs, mapper)
n = torch.arange(a.frames * dabsynth.NB_FRAME_SAMPLES + 2656 + 5000, device=dev, dtype=torch.float64)
x = torch.cat([torch.zeros(5000 + 2656, dtype=torch.complex64, device=dev), f2.reshape(-1).repeat((a.frames + 1) // 2)[:a.frames * dabsynth.NB_FRAME_SAMPLES]])
x = x * torch.polar(torch.ones_like(n), 2 * np.pi * 1.3e-3 * n).to(torch.complex64)
x[:5000] = x[-5000:]
x = x + 0.02 * torch.view_as_complex(torch.randn((x.numel(), 2), device=dev))
sv = torch.view_as_real(x)
peak = float(sv.abs().max().item())
capture = torch.clamp(torch.round(sv / peak * 127.0 + 127.5), 0, 255).to(torch.uint8).cpu().numpy().reshape(-1).tobytes()
del x, n, sv, f2, mux
torch.cuda.synchronize()
torch.cuda.empty_cache()
_free0, _total0 = torch.cuda.mem_get_info()
baseline_device = _total0 - _free0              # what THIS process (torch's context) holds on the device before the receiver starts: subtracted below

cli = a.cli or os.path.join(ROOT, "dab-radio_amd", "host", "apps", "dabgpu_radio_cli")
tmp = tempfile.TemporaryDirectory()
args = [cli, "--configuration", "dab+ofdm", "--ofdm-input-mode", "raw_u8", "--radio-fib-output", os.path.join(tmp.name, "fibs.bin"),
        "--radio-msc-output", os.path.join(tmp.name, "msc_")]
for s in range(a.subchannels):
    args += ["--radio-subchannel", f"{48 * s},48,3,A"]
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = (a.libdir or os.path.join(ROOT, "dab-radio_amd")) + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
p = subprocess.Popen(args, stdin=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
samples = []                                                 # (seconds, repetitions written, VmRSS kB, device bytes used)
written = [0]
stop = threading.Event()


def sample():
    while not stop.is_set():
        try:
            rss = int(re.search(r"VmRSS:\s+(\d+)", open(f"/proc/{p.pid}/status").read()).group(1))
        except Exception:
            break
        free, total = torch.cuda.mem_get_info()
        samples.append((time.perf_counter(), written[0], rss, total - free))
        time.sleep(0.05)


th = threading.Thread(target=sample)
th.start()
t0 = time.perf_counter()
try:
    for r in range(a.repeats):
        p.stdin.write(capture)
        written[0] = r + 1
    p.stdin.close()
except BrokenPipeError:
    pass
err = p.stderr.read().decode()
rc = p.wait()
t1 = time.perf_counter()
stop.set()
th.join()
if rc != 0:
    print(err[-2000:], file=sys.stderr)
    sys.exit(rc)
m = re.search(r"ofdm: frames_read=(\d+) frames_desync=(\d+)", err)
r = re.search(r"radio: frames=(\d+) fibs_crc_ok=(\d+) fib_groups_with_failures=(\d+)(.*)", err)
sub_bytes = [int(v) for v in re.findall(r"subchannel\d+_bytes=(\d+)", r.group(4))]


def at(frac):
    k = min(len(samples) - 1, int(len(samples) * frac))
    return {"repetitions_written": samples[k][1], "host_rss_MB": round(samples[k][2] / 1024, 1), "device_used_MB": round(samples[k][3] / 2**20, 1),
            "receiver_device_MB": round((samples[k][3] - baseline_device) / 2**20, 1)}


frames = int(m.group(1))
out = {"what": __doc__.split("\n\n")[0], "frames_per_repetition": a.frames, "repetitions": a.repeats, "sub_channels": a.subchannels,
       "frames_read": frames, "frames_desync": int(m.group(2)), "radio_frames": int(r.group(1)), "fibs_crc_ok": int(r.group(2)),
       "fib_groups_with_failures": int(r.group(3)), "sub_channel_bytes": sorted(set(sub_bytes)), "seconds": round(t1 - t0, 2),
       "frames_per_s_incl_pipe": round(frames / (t1 - t0), 1), "signal_minutes": round(frames * 0.096 / 60, 1),
       "measuring_process_device_MB": round(baseline_device / 2**20, 1), "memory_at_25_percent": at(0.25), "memory_at_60_percent": at(0.6), "memory_at_end": at(0.92), "samples": len(samples)}
# every repetition re-acquires: at most the first frame (coarse timing) and the cut last one are lost per repetition; the sub-channel
# bytes lag 15 CIFs behind per re-acquisition
out["ok"] = bool(frames >= a.repeats * (a.frames - 2) and out["fibs_crc_ok"] >= 12 * (frames - 2 * a.repeats)
                 and len(set(sub_bytes)) == 1 and out["memory_at_end"]["host_rss_MB"] <= out["memory_at_25_percent"]["host_rss_MB"] + 8
                 and out["memory_at_end"]["device_used_MB"] <= out["memory_at_25_percent"]["device_used_MB"] + 8)
print(json.dumps(out))
sys.exit(0 if out["ok"] else 1)
