"""Generates tests/golden/io_format_vectors.npz: inputs and the outputs produced by the reference's OWN file-format
helpers (examples/app_helpers/app_iq_readers.h, app_wav_reader.h, app_viterbi_convert_block.h compiled in place into
oracle/_ref/libdab_ref.so).  DATA only.  Run from the repo root:  python tests/golden/make_golden_io.py
"""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

GUID_TAIL = bytes([0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71])


def raw_payload(rng, fmt, n_comp):
    """component bytes with the edge values of the type first, random afterwards"""
    size = O.iq_component_bytes(fmt)
    mode = O.IQ_MODES[fmt]
    if size == 1:
        vals = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, n_comp - 256, dtype=np.uint8)])
        return vals.tobytes()
    if "f32" in mode or "f64" in mode:
        dt = np.float32 if "f32" in mode else np.float64
        edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3.4e38, 1e-46, 0.1, 1 / 3], dtype=np.float64)
        if dt is np.float64:
            edge = np.concatenate([edge, np.array([1e300, -1e300, 1e-50, 1e-300, 3.4028235677973366e38, 1.0000000596046448,
                                                   1.00000017881393433, 5e-324, 1.401298464324817e-45, 7.006492321624085e-46])])
        with np.errstate(over="ignore"):
            vals = np.concatenate([edge.astype(dt), (rng.standard_normal(n_comp - edge.size) * 10.0 ** rng.integers(-6, 6)).astype(dt)])
        b = vals.astype(dt.__name__ and ("<" + ("f4" if dt is np.float32 else "f8")))
        return (b.byteswap() if mode.endswith("b") else b).tobytes()
    bits = 8 * size
    edge = np.array([0, 1, 2, (1 << (bits - 1)) - 1, 1 << (bits - 1), (1 << (bits - 1)) + 1, (1 << bits) - 1, (1 << bits) - 2,
                     0x123456 & ((1 << bits) - 1), 0x80FF01 & ((1 << bits) - 1)], dtype=np.uint64)
    vals = np.concatenate([edge, rng.integers(0, 1 << bits, n_comp - edge.size, dtype=np.uint64)])
    out = bytearray()
    be = mode.endswith("b") and mode.startswith("raw")
    for v in vals:
        out += int(v).to_bytes(size, "big" if be else "little")
    return bytes(out)


def wav_image(code, bits, channels, payload, fmt_size=16, extra_chunks=(), rate=2048000, extensible_sub=None):
    align = channels * bits // 8
    fmt = struct.pack("<HHIIHH", 0xFFFE if extensible_sub is not None else code, channels, rate, rate * align, align, bits)
    if fmt_size == 18:
        fmt += struct.pack("<H", 0)
    elif fmt_size == 40:
        fmt += struct.pack("<HHIH", 22, bits, 3, extensible_sub if extensible_sub is not None else code) + GUID_TAIL
    body = b"WAVE" + b"fmt " + struct.pack("<I", fmt_size) + fmt
    eff = extensible_sub if extensible_sub is not None else code
    if eff != 1:
        body += b"fact" + struct.pack("<II", 4, len(payload) // max(align, 1))
    for cid, data in extra_chunks:
        body += cid + struct.pack("<I", len(data)) + data
    body += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def ref_read(R, path, mode, n_samples):
    out = np.zeros(2 * n_samples + 8, np.float32)
    got = R.ref_iq_read_file(path.encode(), mode.encode(), out.ctypes.data, n_samples + 4, 777)
    return got, out[:2 * max(got, 0)].view(np.uint32).copy()


def main():
    R = O.ref()
    assert R is not None, "oracle/_ref/libdab_ref.so missing: needs /root/reference"
    rng = np.random.default_rng(20251002)
    out = {}
    tmp = tempfile.mkdtemp()
    n_comp = 2 * 701                       # odd sample count: exercises the ragged tail of the device kernel
    for fmt in range(14):
        mode = O.IQ_MODES[fmt]
        payload = raw_payload(rng, fmt, n_comp)
        path = os.path.join(tmp, mode)
        open(path, "wb").write(payload)
        got, y = ref_read(R, path, mode, n_comp // 2)
        assert got == n_comp // 2, (mode, got)
        out[f"{mode}_in"] = np.frombuffer(payload, np.uint8)
        out[f"{mode}_out_u32"] = y
    wav_cases = {"wav_pcm8": (1, 8), "wav_pcm16": (1, 16), "wav_pcm24": (1, 24), "wav_pcm32": (1, 32),
                 "wav_f32": (3, 32), "wav_f64": (3, 64), "wav_alaw": (6, 8), "wav_mulaw": (7, 8)}
    headers = []
    variants = [dict(fmt_size=16), dict(fmt_size=18), dict(fmt_size=40),
                dict(fmt_size=16, extra_chunks=((b"LIST", b"INFOISFT\x06\x00\x00\x00dabgpu"), (b"junk", b"\x00" * 11)))]
    for k, (name, (code, bits)) in enumerate(wav_cases.items()):
        fmt = O.IQ_MODES.index(name)
        payload = raw_payload(rng, fmt, n_comp)
        v = dict(variants[k % len(variants)])
        if v["fmt_size"] == 40:
            v["extensible_sub"] = code
        img = wav_image(code, bits, 2, payload, **v)
        path = os.path.join(tmp, name + ".wav")
        open(path, "wb").write(img)
        got, y = ref_read(R, path, "wav", n_comp // 2)
        assert got == n_comp // 2, (name, got)
        h = np.zeros(7, np.uint64)
        assert R.ref_wav_header(path.encode(), h.ctypes.data) == 0
        out[f"{name}_image"] = np.frombuffer(img, np.uint8)
        out[f"{name}_out_u32"] = y
        out[f"{name}_header"] = h
    # header accept / reject table: file images and whether the reference's reader constructor accepts them
    good = wav_image(1, 16, 2, b"\x01\x02\x03\x04" * 4)
    imgs = {
        "ok_pcm16": good,
        "ok_mono": wav_image(1, 16, 1, b"\x01\x02" * 4),
        "ok_ext_float": wav_image(3, 32, 2, b"\x00" * 16, fmt_size=40, extensible_sub=3),
        "bad_riff": b"RIFX" + good[4:],
        "bad_wave": good[:8] + b"WAVX" + good[12:],
        "bad_fmt_id": good[:12] + b"fmtx" + good[16:],
        "bad_fmt_size": good[:16] + struct.pack("<I", 20) + good[20:],
        "bad_code": good[:20] + struct.pack("<H", 2) + good[22:],
        "bad_channels": good[:22] + struct.pack("<H", 3) + good[24:],
        "bad_bits": wav_image(1, 12, 2, b"\x00" * 12),
        "bad_float_bits": wav_image(3, 16, 2, b"\x00" * 8),
        "bad_alaw_bits": wav_image(6, 16, 2, b"\x00" * 8),
        "truncated_header": good[:30],
        "no_data_chunk": good[:36],
        "float_without_fact": wav_image(1, 32, 2, b"\x00" * 16)[:20] + struct.pack("<H", 3) + wav_image(1, 32, 2, b"\x00" * 16)[22:],
        "ext_bad_guid": wav_image(1, 16, 2, b"\x00" * 8, fmt_size=40, extensible_sub=1)[:-(8 + 8 + 14)] + b"\x01" * 14
                        + wav_image(1, 16, 2, b"\x00" * 8, fmt_size=40, extensible_sub=1)[-(8 + 8):],
        "skip_chunk_past_eof": good[:36] + b"LIST" + struct.pack("<I", 1000) + b"abc",
    }
    names, accept = [], []
    for name, img in imgs.items():
        path = os.path.join(tmp, "hdr_" + name)
        open(path, "wb").write(img)
        h = np.zeros(7, np.uint64)
        st = R.ref_wav_header(path.encode(), h.ctypes.data)
        names.append(name); accept.append(st == 0)
        out[f"hdr_{name}_image"] = np.frombuffer(img, np.uint8)
        out[f"hdr_{name}_fields"] = h
    out["hdr_names"] = np.array(names)
    out["hdr_accept"] = np.array(accept)
    # soft/hard bit converters
    b = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 1003, dtype=np.uint8)])
    soft = np.zeros(b.size * 8, np.int8)
    R.ref_bytes_to_bits(b.ctypes.data, b.size, soft.ctypes.data)
    out["hard_in"] = b; out["hard_to_soft"] = soft
    s = np.concatenate([np.array([0, 1, -1, 127, -127, -128, 5, -5], np.int8), rng.integers(-128, 128, 8 * 1258, dtype=np.int8)])
    hb = np.zeros(s.size // 8, np.uint8)
    R.ref_bits_to_bytes(s.ctypes.data, hb.size, hb.ctypes.data)
    out["soft_in"] = s; out["soft_to_hard"] = hb
    path = os.path.join(ROOT, "tests", "golden", "io_format_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", dict(zip(names, accept)))


if __name__ == "__main__":
    main()
