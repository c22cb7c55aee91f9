#!/usr/bin/env python3
"""Instruction histogram of one kernel's ISA (development tool; where bench.py's instructions-per-step constants come from).
    tools/isa_count.py viterbi_lanes.hip vit_lanes_kernelILi0ELi1E            # lists labels, barriers and stores with line numbers
    tools/isa_count.py viterbi_lanes.hip vit_lanes_kernelILi0ELi1E 1618 3370  # histogram of that line range of the kernel
    ... [a-b ...]  further arguments: line ranges to leave out (e.g. the slow-path branch of a guarded block)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(ROOT, "dab-radio_amd", "csrc")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
k = text[start:end]
if len(sys.argv) < 5:
    for i, l in enumerate(k, 1):
        if re.match(r"^\.LBB|\s+s_barrier|\s+s_cbranch|\s+global_store|\s+global_load_dwordx4", l):
            print(i, l.strip()[:100])
    sys.exit(0)
lo, hi = int(sys.argv[3]), int(sys.argv[4])
skip = [tuple(map(int, a.split("-"))) for a in sys.argv[5:]]
c = collections.Counter()
for i, l in enumerate(k[lo - 1:hi], start=lo):
    if any(a <= i <= b for a, b in skip):
        continue
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print("total", sum(c.values()), "VALU", sum(v for n, v in c.items() if n.startswith("v_")), "LDS", sum(v for n, v in c.items() if n.startswith("ds_")),
      "VMEM", sum(v for n, v in c.items() if n.startswith(("global_", "buffer_", "flat_", "scratch_"))))
for n, v in sorted(c.items(), key=lambda x: -x[1]):
    print(f"{v:6d} {n}")
