# development: where the three Viterbi mappings cross for the canonical multiplex (FIC + 18 x 48 CU EEP 3-A per ensemble, one decode call) and for the FIC alone
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dab-radio_amd"), os.path.join(ROOT, "tools")]
import torch, dabgpu, bench
dev = torch.device("cuda", 0)
out = {"msc_decode_call_us": {}, "auto_choice": {}}
for E in [int(a) for a in sys.argv[1:]] or [32, 48, 64, 80, 100, 128, 160]:
    row = {}
    for name, m in (("wave", 1), ("octet", 3), ("lane", 2)):
        ctx = dabgpu.Context(0)
        ctx.viterbi_set_mapping(m)
        p = bench.Pipeline(ctx, dabgpu, torch, dev, E, min(E, 16), seed=7, inflight=1, layout=1, synced=False)
        p.fill()
        p.timed(p.decode, 5)
        row[name] = round(min(p.timed(p.decode, 20) for _ in range(3)) * 1e3, 1)
        if m == 1:
            out["auto_choice"][E] = dabgpu.Context(0).multiplex_mapping(E, p.mux.subchannels(dabgpu))
        del p, ctx
        torch.cuda.empty_cache()
    out["msc_decode_call_us"][E] = row
print(json.dumps(out))
