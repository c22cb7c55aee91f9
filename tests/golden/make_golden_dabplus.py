"""Generates tests/golden/dabplus_vectors.npz from the reference's OWN Reed_Solomon_Decoder and AAC_Frame_Processor
(compiled in place into oracle/_ref/libdab_ref.so).  DATA only.  Run from the repo root:
python tests/golden/make_golden_dabplus.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import dabplus_model as M  # noqa: E402


def main():
    R = O.ref()
    assert R is not None and hasattr(R, "ref_aac_create"), "oracle/_ref/libdab_ref.so missing: needs /root/reference"
    rng = np.random.default_rng(20251003)
    out = {}
    # ---- RS(120,110): 0..11 symbol errors on valid codewords, plus random words ----
    cws = []
    for k in range(400):
        data = rng.integers(0, 256, 110, dtype=np.uint8)
        cw = np.concatenate([data, O.rs120_encode(data)])
        for j in rng.choice(120, k % 12, replace=False):
            cw[j] ^= rng.integers(1, 256)
        cws.append(cw)
    cws += [rng.integers(0, 256, 120, dtype=np.uint8) for _ in range(40)]
    cws = np.stack(cws)
    counts, fixed, positions = [], [], []
    for cw in cws:
        a = cw.copy(); pos = np.full(10, -1, np.int32)
        counts.append(R.ref_rs120_decode(a.ctypes.data, pos.ctypes.data)); fixed.append(a); positions.append(pos)
    out["rs_in"] = cws; out["rs_count"] = np.array(counts, np.int32); out["rs_out"] = np.stack(fixed); out["rs_positions"] = np.stack(positions)
    # ---- super-frame sequences through AAC_Frame_Processor ----
    names = []
    plans = {
        "clean_96": (96, [0] * 4, 0),
        "misaligned_errors_192": (192, [0, 2, 5, 6, 0, 9, 0, 0], 7),
        "desync_24": (24, [0, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 0, 0], 2),
    }
    for name, (n, errs, skip) in plans.items():
        frames = []
        for s, e in enumerate(errs):
            sf, _, _ = M.make_superframe(O, rng, n, dac_rate=s % 2, sbr_flag=(s // 2) % 2, bad_au_crc=(0,) if s == 2 else ())
            sf = M.corrupt(rng, sf, {0: e})
            frames += list(sf.reshape(5, n))
        frames = np.stack(frames[skip:])
        h = C.c_void_p(R.ref_aac_create())
        ev, lens, byts = [], [], []
        for fr in frames:
            o = np.zeros(12, np.int32); al = np.zeros(6, np.int32); ab = np.zeros((6, 1024), np.uint8)
            R.ref_aac_process(h, fr.ctypes.data, n, o.ctypes.data, al.ctypes.data, ab.ctypes.data, 1024)
            ev.append(o); lens.append(al); byts.append(ab)
        buf = C.create_string_buffer(1 << 20)
        R.ref_aac_log(h, buf, 1 << 20)
        out[f"{name}_log"] = np.frombuffer(buf.value, np.uint8).copy()
        R.ref_aac_destroy(h)
        names.append(name)
        out[f"{name}_frames"] = frames; out[f"{name}_n"] = np.int32(n); out[f"{name}_ref"] = np.stack(ev)
        out[f"{name}_au_len"] = np.stack(lens); out[f"{name}_au_bytes"] = np.stack(byts)
    out["seq_names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "dabplus_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; rs counts:", np.unique(out["rs_count"], return_counts=True))


if __name__ == "__main__":
    main()
