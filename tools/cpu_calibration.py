#!/usr/bin/env python3
"""how fast is the oracle PORT next to the REFERENCE's own objects, on the parts of the reference that compile here?

bench.py's `cpu_baseline` times the oracle (oracle/*.c, kind "port") because the reference's hot path as a whole cannot be built in this image
(FFTW3 and vendor/viterbi_decoder are absent).  The translation units that do compile from the reference's own sources are already under
oracle/_ref/libdab_ref.so (oracle/Makefile): this script times them against the oracle's counterparts on the same inputs, one thread, and
writes the ratios -- the calibration to read `cpu_baseline` with.  Runs in the build container only (needs /root/reference for _ref).

    python tools/cpu_calibration.py [--seconds 1.5] [--out profiles/r05/cpu_calibration.json]
"""
import argparse, ctypes as C, json, os, platform, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=1.5, help="time budget per measurement")
ap.add_argument("--out", default=None)
a = ap.parse_args()
R, L = O.ref(), O.lib()
if R is None:
    sys.exit("oracle/_ref/libdab_ref.so is not built: this calibration needs the reference tree (make -C oracle ref)")
rng = np.random.default_rng(1)


def per_call_us(fn, seconds):
    fn(); fn()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(8):
            fn()
        n += 8
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return dt / n * 1e6


def p(arr):
    return arr.ctypes.data_as(C.c_void_p)


rows = []


def row(name, what, unit_per_call, unit, ref_us, port_us, extra=None):
    r = {"piece": name, "what": what, "reference_us_per_call": round(ref_us, 3), "port_us_per_call": round(port_us, 3),
         "reference_rate": unit_per_call / ref_us, "port_rate": unit_per_call / port_us, "rate_unit": unit + " per us",
         "port_time_over_reference_time": round(port_us / ref_us, 3)}
    if extra:
        r.update(extra)
    rows.append(r)
    print(f"{name:28s} reference {ref_us:10.2f} us   port {port_us:10.2f} us   port / reference = {port_us / ref_us:.2f}", file=sys.stderr)


# ---- PLL over one frame buffer: 77 symbols x 2552 samples (ofdm_demodulator.cpp:672-678 -> dsp/apply_pll.cpp) ----
n = 77 * 2552
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
y = np.empty_like(x)
f, dt0 = np.float32(1.3e-3), np.float32(0.21)
L.dab_apply_pll.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
L.dab_apply_pll_scalar.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
t_ref = per_call_us(lambda: R.ref_apply_pll(p(x), p(y), n, f, dt0), a.seconds)
t_base = per_call_us(lambda: R.ref_apply_pll_baseline(p(x), p(y), n, f, dt0), a.seconds)
t_port = per_call_us(lambda: L.dab_apply_pll(p(x), p(y), n, f, dt0), a.seconds)
t_port_s = per_call_us(lambda: L.dab_apply_pll_scalar(p(x), p(y), n, f, dt0), a.seconds)
row("apply_pll (AVX2+FMA build)", "PLL of one frame buffer, 196,504 samples; reference = its default preset's object (-mavx2 -mfma), port = oracle dab_apply_pll "
    "(the same element arithmetic, scalar code, target_clones)", n, "samples", t_ref, t_port)
row("apply_pll (scalar build)", "the same; reference = its baseline object (no AVX), port = oracle dab_apply_pll_scalar", n, "samples", t_base, t_port_s)

# ---- cyclic-prefix correlation: sum of s[2048 + k] * conj(s[k]) over 504 samples x 76 symbols (:768-777 -> dsp/complex_conj_mul_sum.cpp) ----
frame = (rng.standard_normal(76 * 2552) + 1j * rng.standard_normal(76 * 2552)).astype(np.complex64).reshape(76, 2552)
out2 = np.zeros(2, np.float32)


class CF(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


L.dab_cp_correlation.restype = CF
L.dab_cp_correlation.argtypes = [C.c_void_p]
# 504 samples are ~0.2 us of work, a ctypes call ~1.5 us: the 76 calls of a frame are looped in C (helper compiled on the fly)
import subprocess, tempfile
_helper_src = r"""
#include <stddef.h>
typedef struct { float re, im; } cf;
void loop_ref(void (*fn)(const float*, const float*, size_t, float*), const float* frame, int reps, float* out) {
    for (int r = 0; r < reps; r++) for (int i = 0; i < 76; i++) fn(frame + (size_t)i * 2552 * 2 + 2048 * 2, frame + (size_t)i * 2552 * 2, 504, out);
}
void loop_port(cf (*fn)(const void*), const float* frame, int reps, float* out) {
    for (int r = 0; r < reps; r++) for (int i = 0; i < 76; i++) { cf v = fn(frame + (size_t)i * 2552 * 2); out[0] = v.re; out[1] = v.im; }
}
"""
_tmp = tempfile.mkdtemp()
open(os.path.join(_tmp, "loop.c"), "w").write(_helper_src)
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", os.path.join(_tmp, "loop.c"), "-o", os.path.join(_tmp, "loop.so")], check=True)
H = C.CDLL(os.path.join(_tmp, "loop.so"))
H.loop_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
H.loop_port.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
REPS = 20
fp = lambda f: C.cast(f, C.c_void_p)
t_ref = per_call_us(lambda: H.loop_ref(fp(R.ref_conj_mul_sum), p(frame), REPS, p(out2)), a.seconds) / REPS
t_base = per_call_us(lambda: H.loop_ref(fp(R.ref_conj_mul_sum_baseline), p(frame), REPS, p(out2)), a.seconds) / REPS
t_port = per_call_us(lambda: H.loop_port(fp(L.dab_cp_correlation), p(frame), REPS, p(out2)), a.seconds) / REPS
row("complex_conj_mul_sum (AVX2)", "cyclic-prefix correlations of one frame: 76 calls x 504 samples, looped in C; port = oracle dab_cp_correlation (a fixed reduction "
    "tree, DESIGN 3.3, instead of the reference's 4 strided partial sums)", 76 * 504, "samples", t_ref, t_port)
row("complex_conj_mul_sum (scalar)", "the same against the reference's baseline object (sequential sum)", 76 * 504, "samples", t_base, t_port)

# ---- time de-interleaver: Consume + Deinterleave of a 48 CU sub-channel and of a whole CIF (cif_deinterleaver.cpp:13-71) ----
for cu in (48, 864):
    nb = cu * 8
    bits = rng.integers(-127, 128, nb * 8).astype(np.int8)
    out = np.empty(nb * 8, np.int8)
    h = C.c_void_p(R.ref_deint_create(nb))
    L.dab_deinterleaver_create.restype = C.c_void_p
    ho = C.c_void_p(L.dab_deinterleaver_create(nb))
    L.dab_deinterleaver_consume.argtypes = [C.c_void_p, C.c_void_p]
    L.dab_deinterleaver_deinterleave.argtypes = [C.c_void_p, C.c_void_p]
    for _ in range(16):
        R.ref_deint_consume(h, p(bits), bits.size); L.dab_deinterleaver_consume(ho, p(bits))

    def ref_d():
        R.ref_deint_consume(h, p(bits), bits.size); R.ref_deint_deinterleave(h, p(out), out.size)

    def port_d():
        L.dab_deinterleaver_consume(ho, p(bits)); L.dab_deinterleaver_deinterleave(ho, p(out))
    row(f"CIF_Deinterleaver ({cu} CU)", f"Consume + Deinterleave of one CIF's {cu * 64} soft bits", cu * 64, "soft bits", per_call_us(ref_d, a.seconds), per_call_us(port_d, a.seconds))
    R.ref_deint_destroy(h)

# ---- RS(120,110) with 0 and with 5 symbol errors (reed_solomon_decoder.cpp through aac_frame_processor.cpp:20-26) ----
if hasattr(R, "ref_rs120_decode"):
    data = rng.integers(0, 256, 110).astype(np.uint8)
    cw = np.concatenate([data, O.rs120_encode(data)])
    pos = np.zeros(16, np.int32)
    L.dab_rs120_decode.argtypes = [C.c_void_p, C.c_void_p]
    for n_err in (0, 5):
        bad = cw.copy()
        bad[rng.choice(120, n_err, replace=False)] ^= 0x5A
        wr, wo = bad.copy(), bad.copy()

        def ref_rs():
            wr[:] = bad; return R.ref_rs120_decode(p(wr), p(pos))

        def port_rs():
            wo[:] = bad; return L.dab_rs120_decode(p(wo), p(pos))
        assert ref_rs() == port_rs() == n_err and np.array_equal(wr, cw) and np.array_equal(wo, cw)
        row(f"Reed_Solomon_Decoder ({n_err} errors)", f"one RS(120,110) code word with {n_err} corrupted symbols (incl. the 120-byte restore of the input)", 120, "code word bytes",
            per_call_us(ref_rs, a.seconds), per_call_us(port_rs, a.seconds))

geo = float(np.exp(np.mean([np.log(r["port_time_over_reference_time"]) for r in rows])))
ofdm = [r for r in rows if r["piece"].startswith(("apply_pll (AVX2", "complex_conj_mul_sum (AVX2"))]
doc = {"what": "one-thread time of the oracle port divided by the time of the reference's own compiled objects on the same inputs, for the pieces of the "
               "reference that build in this image from its own sources (oracle/Makefile -> oracle/_ref); > 1 = the port is slower than the reference",
       "cpu": platform.processor() or platform.machine(), "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
       "threads": 1, "seconds_per_measurement": a.seconds, "rows": rows, "geometric_mean_port_over_reference": round(geo, 3),
       "ofdm_pieces_avx2_port_over_reference": [r["port_time_over_reference_time"] for r in ofdm],
       "not_calibrated": "FFT (FFTW3 absent), DQPSK / soft-bit stage and the synchroniser (same translation unit as <fftw3.h>), Viterbi ACS / chain-back "
                         "(vendor/viterbi_decoder is an empty submodule): most of the CPU time of the full chain; for those the port IS the only CPU number there is",
       "how_to_read_cpu_baseline": "bench.py's cpu_baseline / cpu_baseline_full time the port.  Where the reference vectorises, the port does too since round 5 -- the PLL "
                                   "(first row: at the speed of the reference's AVX2 object), the 2048-point transform, the DQPSK / soft-bit stage and the Viterbi "
                                   "add-compare-select (AVX2 forms of the oracle's scalar statements, bit-identical, tests/test_oracle_pins.py / test_oracle_properties.py): "
                                   "one frame demodulates in ~1.5 ms and the whole receive chain with 18 sub-channels takes ~4.7 ms on one core of the build container "
                                   "(round 4: ~7.7 ms and ~13 ms).  FFTW's transform and the upstream SIMD Viterbi core may still be faster than these; the GPU / CPU "
                                   "ratio is a reported baseline, not the target."}
print(json.dumps(doc, indent=1))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
