"""Experiment (round 5): does de-phasing the trellis kernel's two halves pay?  The forward pass of vit_lanes_kernel runs at the VALU issue limit,
its chain-back is an HBM-bound burst (3.8 GB of decision words read back by all wavefronts at once).  Two half batches on two streams, the second
started `delay` later: the first half's chain-back then overlaps the second half's forward pass."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/dab-radio_amd", ROOT + "/tools"]
import torch, dabgpu, bench
dev = torch.device("cuda", 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = dabgpu.Context(0)
p = bench.Pipeline(ctx, dabgpu, torch, dev, E, 64, seed=7, inflight=2, layout=1, synced=False)
p.fill()
H, stride, nb = p.H, p.stride, p.cif_bytes
def msc(k, lo, hi, slot):
    p.ctxs[k].msc_decode_frames(p.hist[lo:hi], hi - lo, stride, H, slot, p.subs, p.msc_out[k][lo:hi], 4 * nb, p.msc_res[k][lo * 4 * p.n_sub:hi * 4 * p.n_sub],
                                stream=p.streams[k].cuda_stream, bits_layout=p.layout)
def timeit(fn, reps=6):
    torch.cuda.synchronize(); fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps): fn(r % H)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
out = {}
out["full_one_call"] = timeit(lambda s: msc(0, 0, E, s))
out["two_halves_one_stream"] = timeit(lambda s: (msc(0, 0, E // 2, s), msc(0, E // 2, E, s)))
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
cyc_per_ms = 20_000_000 / ((time.perf_counter() - t0) * 1e3)
out["sleep_cycles_per_ms"] = cyc_per_ms
for delay_ms in (0.0, 0.4, 0.7, 1.0, 1.3, 1.6):
    def two(s, d=delay_ms):
        ev = torch.cuda.Event(); ev.record(p.streams[0]); p.streams[1].wait_event(ev)
        msc(0, 0, E // 2, s)
        with torch.cuda.stream(p.streams[1]):
            if d > 0: torch.cuda._sleep(int(d * cyc_per_ms))
        msc(1, E // 2, E, s)
        ev2 = torch.cuda.Event(); ev2.record(p.streams[1]); p.streams[0].wait_event(ev2)
    out[f"two_streams_delay_{delay_ms}"] = timeit(two)
print(json.dumps(out, indent=1))
