"""Inputs of tests/golden/decoder_vectors.npz, regenerated wherever they are needed from integer arithmetic only (a splitmix64 counter hash in
numpy uint64: no dependence on numpy's Generator streams), with their SHA-256 in the fixture so that a different regeneration fails loudly.

MSC: every protection profile the reference's tables hold -- EEP 1-A..4-A, 1-B..4-B (incl. the n = 1 special case of 2-A and the smallest 4-A) and
all 64 UEP table rows (subchannel_protection_tables.h:21-139) -- as sub-channels packed into multiplexes of <= 864 CU, 20 CIFs each: random payload,
channel coded and time interleaved by the oracle's TRANSMIT side, scaled, noisy (three gain / noise settings, one of them coarsely quantised so that
exact metric ties occur), capacity units no sub-channel occupies filled with noise.  FIC: 48 FIB groups (clean, noisy, noise only, erased)."""
import hashlib

import numpy as np

N_CIFS = 20
N_FIC_GROUPS = 48
SETTINGS = [(100, 40, 1), (80, 42, 1), (80, 60, 16)]          # (gain, noise scale, quantisation step): the third collapses the soft values onto multiples of 16


def splitmix64(seed, n):
    with np.errstate(over="ignore"):
        z = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rand_bytes(seed, n):
    return (splitmix64(seed, n) >> np.uint64(24)).astype(np.uint8)


def noise(seed, n, scale):
    """roughly Gaussian integers: (sum of four hash bytes - 510) * scale >> 7  (standard deviation ~ 1.15 * scale)"""
    h = splitmix64(seed, n)
    s = sum(((h >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.int64) for k in range(4)) - 510
    return (s * scale) >> 7


def soften(bits01, seed, gain, scale, step):
    y = (2 * bits01.astype(np.int64) - 1) * gain + noise(seed, bits01.size, scale).reshape(bits01.shape)
    if step > 1:
        y = ((y + step // 2) // step) * step
    return np.clip(y, -127, 127).astype(np.int8)


def profiles(O):
    """[(name, SubChannel without its address)]"""
    out = []
    for level, type_b, n in ((0, 0, 2), (1, 0, 3), (1, 0, 1), (2, 0, 8), (2, 0, 3), (3, 0, 5), (3, 0, 1), (0, 1, 1), (1, 1, 2), (2, 1, 1), (3, 1, 2)):
        mult = ((12, 8, 6, 4), (27, 21, 18, 15))[type_b][level]
        out.append(("eep_%d%s_n%d" % (level + 1, "AB"[type_b], n), dict(length=mult * n, is_uep=0, uep_index=0, eep_level=level, eep_type=type_b)))
    for row in range(64):
        length = uep_size(O, row)
        out.append(("uep_%02d" % row, dict(length=length, is_uep=1, uep_index=row, eep_level=0, eep_type=0)))
    return out


_UEP_SIZES = None


def uep_size(O, row):
    """sub-channel size of UEP table row `row`, from the plans the reference's own table produced (tests/golden/reference_vectors.npz:subchannel_plans,
    made by tests/golden/make_golden.py from subchannel_protection_tables.h:21-86)"""
    global _UEP_SIZES
    if _UEP_SIZES is None:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))
        _UEP_SIZES = {int(r[2]): int(r[0]) for r in z["subchannel_plans"] if int(r[1]) == 1}
        assert sorted(_UEP_SIZES) == list(range(64))
    return _UEP_SIZES[row]


def cases(O):
    """every profile with its multiplex number, start address and noise setting: [dict(name, mux, start, length, ..., setting)]"""
    out, mux, at = [], 0, 0
    for k, (name, p) in enumerate(profiles(O)):
        if at + p["length"] > 864:
            mux, at = mux + 1, 0
        out.append(dict(name=name, index=k, mux=mux, start=at, setting=k % len(SETTINGS), **p))
        at += p["length"]
    return out


def subchannel(O, c):
    return O.subchannel(c["start"], c["length"], eep_level=c["eep_level"], eep_type=c["eep_type"], is_uep=bool(c["is_uep"]), uep_index=c["uep_index"])


def msc_multiplexes(O):
    """-> (cases, cifs [n_mux][N_CIFS][55296] int8, payload per case [N_CIFS][nbytes])"""
    cs = cases(O)
    n_mux = cs[-1]["mux"] + 1
    cifs = np.stack([soften(np.zeros((N_CIFS, 55296), np.uint8), 9000 + m, 0, 45, 1) for m in range(n_mux)])        # unoccupied capacity: noise
    payload = []
    for c in cs:
        sc = subchannel(O, c)
        pi, lx, nb = O.subchannel_plan(sc)
        pay = rand_bytes(100 + c["index"], N_CIFS * nb).reshape(N_CIFS, nb)
        if sum(4 * int(l) * (8 + int(p_)) for p_, l in zip(pi, lx)) + 12 > c["length"] * 64:
            # UEP table row 34 as the reference lists it (its size field and row 33's are exchanged, subchannel_protection_tables.h:55-56): the code
            # word does not fit into the sub-channel, nothing can be transmitted in it -- the decoders are handed noise and run out of symbols
            # inside the third update() (dab_viterbi_decoder.cpp:157-160)
            payload.append(None)
            continue
        logical = np.stack([O.msc_encode_logical(sc, pay[t]) for t in range(N_CIFS)])
        tx = O.time_interleave(logical)                               # CIF s carries bit i of the logical frame bitrev4(i mod 16) CIFs older (zero before the start)
        gain, scale, step = SETTINGS[c["setting"]]
        cifs[c["mux"], :, c["start"] * 64:(c["start"] + c["length"]) * 64] = soften(tx, 500 + c["index"], gain, scale, step)
        payload.append(pay)
    return cs, cifs, payload


def fic_groups(O):
    """-> (soft [N_FIC_GROUPS][2304] int8, fib data [N_FIC_GROUPS][90])"""
    data = rand_bytes(77, N_FIC_GROUPS * 90).reshape(N_FIC_GROUPS, 90)
    soft = np.zeros((N_FIC_GROUPS, 2304), np.int8)
    for g in range(N_FIC_GROUPS):
        tx = O.fic_encode_group(data[g])
        kind = g % 8
        if kind == 6:
            soft[g] = soften(np.zeros(2304, np.uint8), 7000 + g, 0, 60, 1)                 # noise only: no FIB passes
        elif kind == 7:
            soft[g] = soften(tx, 7000 + g, 127, 0, 1)
            soft[g, 300 + 40 * g:900 + 40 * g] = 0                                         # an erased stretch
        else:
            gain, scale, step = ((127, 0, 1), (100, 40, 1), (60, 50, 1), (80, 60, 16), (45, 50, 1), (40, 40, 8))[kind]
            soft[g] = soften(tx, 7000 + g, gain, scale, step)
    return soft, data


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def oracle_fic(O, soft, model):
    """the oracle's composition for the FIC vectors: per group (number of valid FIBs, their 30 data bytes back to back zero-padded to 90)"""
    n, out = np.zeros(len(soft), np.int32), np.zeros((len(soft), 90), np.uint8)
    for g, s in enumerate(soft):
        eb, em, _ = O.fic_decode_group(s, model)
        k = 0
        for i in range(3):
            if em & (1 << i):
                out[g, 30 * k:30 * k + 30] = eb[32 * i:32 * i + 30]
                k += 1
        n[g] = k
    return n, out


def oracle_msc(O, c, cifs, model):
    """the oracle's composition for one sub-channel: [(cif, bytes)] for every CIF from the 16th on (Deinterleaver + msc_decode_logical)"""
    sc = subchannel(O, c)
    di = O.Deinterleaver(c["length"] * 8)
    out = []
    for t in range(cifs.shape[0]):
        di.consume(cifs[t, c["start"] * 64:(c["start"] + c["length"]) * 64])
        lf = di.deinterleave()
        out.append(None if lf is None else O.msc_decode_logical(sc, lf, model)[0])
    return out
