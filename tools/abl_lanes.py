#!/usr/bin/env python3
"""Ablation of vit_lanes_kernel (the MSC trellis kernel: 69 % of the configs[3] / configs[4] step).  Development tool.

The product source carries no switches: this script DERIVES timing-only variants of dab-radio_amd/csrc/viterbi_lanes.hip by text substitution
(each variant removes one part of the per-step instruction stream and so decodes garbage), builds build/exp/libdabgpu_<tag>.so from each
(the product's other objects), counts the VALU instructions of the forward loop of every variant (the 6-step main loop of
vit_lanes_kernel<0, 1, 5>, renormalisation slow paths left out) and, with --run on a GPU box, times the MSC decode of 4096 ensembles of the
canonical multiplex with each.

    python tools/abl_lanes.py --build                 # here (hipcc cross-compiles): variants + instruction counts -> build/exp/abl_lanes.json
    python tools/abl_lanes.py --run [--ensembles 4096] # on the GPU box: times them, prints the table as JSON
"""
import argparse, collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dab-radio_amd", "csrc")
OUT = os.path.join(ROOT, "build", "exp")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]

GATHER = "    vl_gather<(Q + 1) % 6>(D, w0, w1);\n"
FOLD = ("    { uint32_t a_ = as_u32(D[0]), b_ = as_u32(D[1]);\n"
        "      _Pragma(\"unroll\") for (int r_ = 2; r_ < 32; r_ += 4) { a_ = a_ ^ as_u32(D[r_]) ^ as_u32(D[r_ + 1]); b_ = b_ ^ as_u32(D[r_ + 2]) ^ as_u32(D[r_ + 3]); }\n"
        "      w0 = a_; w1 = b_; }\n")
NODEC = "    w0 = as_u32(N[0]) ^ as_u32(N[17]); w1 = as_u32(N[5]) ^ as_u32(N[30]);\n"
COST_A = "        vl_cost_table<vl_flip(Q), true>(ysym, WF, WX, k508, C);\n"
COST_B = ("        _Pragma(\"unroll\") for (int s_ = 0; s_ < 8; s_++) C[s_] = as_s2(ysym + 0x00010001u * (uint32_t)s_ + (uint32_t)k508);\n")
RENORM_N, RENORM_M = "        vl_renorm(N, total);                                                                          \\\n", \
                     "        vl_renorm(M, total);                                                                          \\\n"

VARIANTS = {
    "base": [],
    "gather_as_xor_fold": [(GATHER, FOLD)],                       # the 30 gather instructions replaced by 15 v_xor3 that keep every D alive
    "no_decisions": [(GATHER, NODEC)],                            # no gather AND no saturated differences (D is dead): add + min only
    "no_cost_table": [(COST_A, COST_B)],                          # branch costs from one add per pattern instead of dots + mads
    "no_renorm": [(RENORM_N, ""), (RENORM_M, "")],                # no threshold test / renormalisation
    "acs_add_min_only": [(GATHER, NODEC), (COST_A, COST_B), (RENORM_N, ""), (RENORM_M, "")],
    # some groups start k x n_steps cycles late: their forward pass then runs beside the others' chain-back (which groups share a SIMD is the
    # hardware's business: bit `sh` of the group number picks the late ones)
    **{f"stagger_b{sh}_{k}": [("    const bool valid = lane < (int)Gd.count;\n",
                         "    if ((group >> %d) & 1) { const unsigned long long t_end_ = __builtin_readcyclecounter() + (unsigned long long)T * %dull;\n"
                         "                     while (__builtin_readcyclecounter() < t_end_) __builtin_amdgcn_s_sleep(64); }\n"
                         "    const bool valid = lane < (int)Gd.count;\n" % (sh, k))] for sh in (3, 5, 8, 10) for k in (600, 1200)},
    "no_chainback": [("    for (int th = T - 1; th >= 6; th -= CB) {\n", "    for (int th = T - 1; th >= 6 && T < 0; th -= CB) {\n")],     # forward pass only
    "no_decision_store": [("        __builtin_nontemporal_store(u4v{wv.x, wv.y, wv.z, wv.w},                                      \\\n"
                           "                                    reinterpret_cast<u4v*>(reinterpret_cast<char*>(grp_dec + (size_t)((TT) >> 1) * 256) + lane16)); \\\n",
                           "        asm volatile(\"\" :: \"v\"(wv.x), \"v\"(wv.y), \"v\"(wv.z), \"v\"(wv.w));                     \\\n")],
}


def make_variant(tag):
    src = open(os.path.join(CSRC, "viterbi_lanes.hip")).read()
    for old, new in VARIANTS[tag]:
        assert src.count(old) >= 1, (tag, old)
        src = src.replace(old, new)
    os.makedirs(os.path.join(OUT, "src"), exist_ok=True)
    path = os.path.join(OUT, "src", f"viterbi_lanes_{tag}.hip")
    open(path, "w").write(src)
    return path


def count_valu(path):
    asm = os.path.join(OUT, "src", os.path.basename(path) + ".s")
    subprocess.run(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", path, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN6dabgpu16vit_lanes_kernelILi0ELi1ELi5E\w*:", l))
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
    k = text[start:end]
    # the main loop = the innermost loop that holds three 16-byte decision stores
    stores = [i for i, l in enumerate(k) if "global_store_dwordx4" in l]
    best = None
    for i, l in enumerate(k):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)
        if not m:
            continue
        # the loop's back edge: last branch to this label
        back = max((j for j, x in enumerate(k) if re.search(r"s_c?branch\w*\s+" + re.escape(m.group(1)) + r"\b", x)), default=None)
        if back is None or back < i:
            continue
        n_st = sum(1 for s_ in stores if i < s_ < back)
        if n_st == 3 and (best is None or back - i < best[1] - best[0]):
            best = (i, back)
    if not best:                                  # (a variant without the decision stores: the loop is not recognised; timing only)
        return {"valu_per_step": float("nan"), "split_per_step": {}, "total_per_step": float("nan")}
    lo, hi = best
    # leave out the renormalisation slow paths: blocks entered by s_cbranch_execz over > 40 instructions
    c = collections.Counter()
    skip_until = None
    for i in range(lo, hi + 1):
        l = k[i].strip()
        if skip_until is not None:
            if l.startswith(skip_until + ":"):
                skip_until = None
            continue
        m = re.match(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = next((j for j in range(i, hi + 1) if k[j].startswith(m.group(1) + ":")), None)
            if tgt is not None and tgt - i > 40:
                skip_until = m.group(1)
            c["s_cbranch_execz"] += 1
            continue
        if not l or l.startswith((".", ";")) or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    valu = sum(v for n, v in c.items() if n.startswith("v_"))
    split = {"acs_add_min_sub": sum(c[n] for n in ("v_pk_add_u16", "v_pk_min_i16", "v_pk_sub_i16")),
             "gather_perm_bfi_xor": c["v_perm_b32"] + c["v_bfi_b32"] + c["v_xor3_b32"] + c["v_xor_b32"],
             "cost_dot_mad": c["v_dot4_i32_i8"] + c["v_pk_mad_i16"],
             "s_nop": c["s_nop"]}
    split["other_valu"] = valu - split["acs_add_min_sub"] - split["gather_perm_bfi_xor"] - split["cost_dot_mad"]
    return {"valu_per_step": valu / 6.0, "split_per_step": {n: v / 6.0 for n, v in split.items()}, "total_per_step": sum(c.values()) / 6.0}


def build(tag, path):
    obj = os.path.join(OUT, f"viterbi_lanes_{tag}.o")
    subprocess.run(["hipcc"] + FLAGS + ["-c", path, "-o", obj], check=True, stderr=subprocess.DEVNULL)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "viterbi_lanes.o"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libdabgpu_abl_{tag}.so"), obj] + others, check=True)


def run_one(tag, ensembles):
    env = dict(os.environ, DABGPU_LIB=os.path.join(OUT, f"libdabgpu_abl_{tag}.so"))
    code = f"""
import sys, json
sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'dab-radio_amd')!r}, {os.path.join(ROOT, 'tools')!r}]
import torch, dabgpu, bench
ctx = dabgpu.Context(0)
p = bench.Pipeline(ctx, dabgpu, torch, torch.device('cuda', 0), {ensembles}, 64, seed=7, inflight=1, layout=1, synced=False)
p.fill()
p.timed(p.msc, 3)
t = [p.timed(p.msc, 6) for _ in range(3)]
print(json.dumps({{"msc_call_ms": min(t), "msc_call_ms_runs": t}}))
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    if res.returncode != 0:
        return {"error": res.stderr[-500:]}
    return json.loads(res.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--ensembles", type=int, default=4096)
    a = ap.parse_args()
    meta_path = os.path.join(OUT, "abl_lanes.json")
    if a.build:
        meta = {}
        only = os.environ.get("ABL_ONLY")
        if only and os.path.exists(meta_path):
            meta = json.load(open(meta_path))
        for tag in VARIANTS:
            if only and tag not in only.split(","):
                continue
            path = make_variant(tag)
            meta[tag] = count_valu(path)
            build(tag, path)
            print(tag, json.dumps(meta[tag]), file=sys.stderr)
        json.dump(meta, open(meta_path, "w"), indent=1)
    if a.run:
        meta = json.load(open(meta_path))
        only = os.environ.get("ABL_ONLY")
        if only:
            meta = {t: m for t, m in meta.items() if t == "base" or t in only.split(",")}
        for tag in meta:
            meta[tag].update(run_one(tag, a.ensembles))
        base = meta["base"]
        for tag, m in meta.items():
            if "msc_call_ms" in m:
                m["ms_vs_base"] = m["msc_call_ms"] / base["msc_call_ms"]
                m["valu_vs_base"] = m["valu_per_step"] / base["valu_per_step"]
        print(json.dumps({"ensembles": a.ensembles, "what": "dabgpu_msc_decode_frames_layout (gather + trellis, 18 x 48 CU EEP 3-A), timing-only variants of vit_lanes_kernel",
                          "variants": meta}))


if __name__ == "__main__":
    main()
