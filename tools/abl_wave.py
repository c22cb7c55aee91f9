#!/usr/bin/env python3
"""Ablation of viterbi_kernel (one wavefront per codeword: what ONE receiver's frame is decoded by -- 4 FIB groups + 72 sub-channel codewords in one launch).
Development tool, same method as tools/abl_lanes.py: timing-only variants DERIVED from dab-radio_amd/csrc/viterbi.hip by text substitution (each removes
one part and so decodes garbage), built into build/exp/libdabgpu_ablw_<tag>.so, timed on the FIC + MSC decode call of one ensemble.

    python tools/abl_wave.py --build      # here
    python tools/abl_wave.py --run        # on the GPU box -> JSON
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dab-radio_amd", "csrc")
OUT = os.path.join(ROOT, "build", "exp")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
FWD = "            forward_block<TIE>(metric, hist, renorm_total, K, ypk, t0, n_steps, lane, my_dec16);\n"
CB = "        for (int tb = ((n_steps - 1) / VBLOCK) * VBLOCK; tb >= 0; tb -= VBLOCK) {\n"
LOAD = "                if (step < n_steps && ((fkeep >> q) & 1)) y = src_base[age_off[fidx[q] & 15] + (unsigned long long)((unsigned)fidx[q] >> ish)];\n"
ACS_Z = """    // own predecessor at branch cost e = 508 - dot, partner at 1016 - e = 508 + dot; the dot does not depend on the metrics
    const uint32_t ds = (uint32_t)__builtin_amdgcn_sdot4(K.sig[P], ysym, 0, false);
    const uint32_t m508 = metric + 508u;
    const uint32_t c_self = m508 - ds;
    const uint32_t c_part = xchg<P>(m508, lane) + ds;                    // (the DPP phases: one v_add_u32_dpp)
"""
# two dots that accumulate on the metrics in place (the form of rounds 1-4, with the 16-bit minimum): 9 instructions, a chain of 5
ACS_X = """    const uint32_t m508 = metric + 508u;
    const uint32_t p508 = xchg<P>(m508, lane);
    const uint32_t c_self = (uint32_t)__builtin_amdgcn_sdot4(~K.sig[P] + 0x01010101, ysym, (int)m508, false);
    const uint32_t c_part = (uint32_t)__builtin_amdgcn_sdot4(K.sig[P], ysym, (int)p508, false);
"""
# both branch costs ready before the metric: 11 instructions, a chain of 2
ACS_Y = """    const uint32_t c_hi = (uint32_t)__builtin_amdgcn_sdot4(K.sig[P], ysym, 508, false);
    const uint32_t c_lo = 1016u - c_hi;
    const uint32_t c_self = metric + c_lo;
    const uint32_t c_part = xchg<P>(metric, lane) + c_hi;
"""
VARIANTS = {
    "base": [],
    "acs_two_dots_in_place": [(ACS_Z, ACS_X)],
    "acs_costs_before_metric": [(ACS_Z, ACS_Y)],
    "symbols_from_vgpr": [("    const int ysym = __builtin_amdgcn_readlane(ypk[J >> 4], 4 * (J & 15));\n", "    const int ysym = ypk[J >> 4] + J;\n")],
    "no_decision_bits": [("    hist = __builtin_amdgcn_alignbit(hist, (uint32_t)d, 31);          // hist = hist << 1 | sign(d)\n", "")],
    "symbols_from_vgpr_no_decision_bits": [("    const int ysym = __builtin_amdgcn_readlane(ypk[J >> 4], 4 * (J & 15));\n", "    const int ysym = ypk[J >> 4] + J;\n"),
                                           ("    hist = __builtin_amdgcn_alignbit(hist, (uint32_t)d, 31);          // hist = hist << 1 | sign(d)\n", "")],
    "no_forward": [(FWD, '            asm volatile("" :: "v"(ypk[0]), "v"(ypk[1]), "v"(ypk[2]));\n')],
    "no_chainback": [(CB, "        for (int tb = ((n_steps - 1) / VBLOCK) * VBLOCK; tb >= 0 && n_steps < 0; tb -= VBLOCK) {\n")],
    "no_symbol_loads": [(LOAD, "                if (step < n_steps && ((fkeep >> q) & 1)) y = (int)(fidx[q] & 63) - 32;\n")],
    "no_forward_no_chainback": [(FWD, '            asm volatile("" :: "v"(ypk[0]), "v"(ypk[1]), "v"(ypk[2]));\n'),
                                (CB, "        for (int tb = ((n_steps - 1) / VBLOCK) * VBLOCK; tb >= 0 && n_steps < 0; tb -= VBLOCK) {\n")],
}


def build(tag):
    src = open(os.path.join(CSRC, "viterbi.hip")).read()
    for old, new in VARIANTS[tag]:
        assert src.count(old) == 1, (tag, old)
        src = src.replace(old, new)
    os.makedirs(os.path.join(OUT, "src"), exist_ok=True)
    path = os.path.join(OUT, "src", f"viterbi_{tag}.hip")
    open(path, "w").write(src)
    obj = os.path.join(OUT, f"viterbi_{tag}.o")
    subprocess.run(["hipcc"] + FLAGS + ["-c", path, "-o", obj], check=True, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "viterbi.o"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libdabgpu_ablw_{tag}.so"), obj] + others, check=True)


def run_one(tag, ensembles):
    env = dict(os.environ, DABGPU_LIB=os.path.join(OUT, f"libdabgpu_ablw_{tag}.so"))
    code = f"""
import sys, json
sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'dab-radio_amd')!r}, {os.path.join(ROOT, 'tools')!r}]
import torch, dabgpu, bench
ctx = dabgpu.Context(0)
p = bench.Pipeline(ctx, dabgpu, torch, torch.device('cuda', 0), {ensembles}, 1, seed=7, inflight=1, layout=0, synced=False)
p.fill()
p.timed(p.decode, 20)
t = [p.timed(p.decode, 100) for _ in range(3)]
print(json.dumps({{"decode_call_us": min(t) * 1e3, "runs_us": [x * 1e3 for x in t]}}))
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    return json.loads(res.stdout.strip().splitlines()[-1]) if res.returncode == 0 else {"error": res.stderr[-600:]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--ensembles", type=int, default=1)
    a = ap.parse_args()
    if a.build:
        for tag in VARIANTS:
            build(tag)
    if a.run:
        out = {tag: run_one(tag, a.ensembles) for tag in VARIANTS}
        print(json.dumps({"what": "dabgpu_decode_frames_layout, %d ensemble(s) of 18 x 48 CU EEP 3-A + FIC, wave mapping; timing-only variants of viterbi_kernel" % a.ensembles, "variants": out}))
