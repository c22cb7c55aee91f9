// ofdm_device.h -- device-side arithmetic shared by the OFDM kernels (ofdm_demod.hip, ofdm_sync.hip).
// Every function here is part of the arithmetic contract of DESIGN.md section 3 and mirrors, operation for
// operation, the CPU oracle's restatement; compile with -ffp-contract=off -fno-slp-vectorize.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabgpu_host_logic.h"

namespace dabgpu {

typedef float f4 __attribute__((ext_vector_type(4)));
struct alignas(8) f2 { float x, y; };
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }

constexpr int NB_SYMBOL_PERIOD = 2552;
constexpr int NB_FFT = 2048;
constexpr int NB_CP = 504;
constexpr int NB_FRAME_SAMPLES = 196608;
constexpr int NB_SYM_BITS = 3072;
constexpr int NB_FRAME_BITS = 230400;
constexpr int NB_FRAME_SYMBOLS = 76;
constexpr int NB_FIC_SYMBOLS = 3;        // data symbols 1..3 carry the FIC (9216 soft bits), 4..75 the four CIFs of 18 symbols each
constexpr int NB_CIF_SYMBOLS = 18;
constexpr int NB_CIF_BITS = 55296;
constexpr int WAVE_PATCH = 576;          // float2 elements per wave transpose patch (8 rows x 72, padded)

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return f2{fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 mk2(float a, float b) { return f2{a, b}; }

// ---- PLL: chebyshev sine on a (cos-arg, sin-arg) pair, FMA Horner (chebyshev_sine.h:82-107, __FMA__) ----
__device__ __forceinline__ f2 cheb2(f2 x) {
    const f2 z = x * x;
    f2 b = fma2(mk2(3.20396066f, 3.20396066f), z, mk2(-14.07150173f, -14.07150173f));
    b = fma2(b, z, mk2(38.50016403f, 38.50016403f));
    b = fma2(b, z, mk2(-67.07687378f, -67.07687378f));
    b = fma2(b, z, mk2(64.83583069f, 64.83583069f));
    b = fma2(b, z, mk2(-25.13274193f, -25.13274193f));
    const f2 c0 = z - mk2(0.25f, 0.25f);
    return (b * c0) * x;
}

// one sample: base = dt0 + float(i4)*f (group of 4), step = (k*f + .25, k*f); v * (cos + j sin)
__device__ __forceinline__ f2 pll1(f2 v, float base, f2 step) {
    f2 d = mk2(base, base) + step;
    d = d - mk2(__builtin_rintf(d.x), __builtin_rintf(d.y));
    const f2 cs = cheb2(d);                               // (cos, sin)
    // c32_mul_avx with FMA (x86/c32_mul.h:9-38): b0 = s*(xi, xr); y = (fma(c,xr,-b0.x), fma(c,xi,+b0.y))
    const float b0x = cs.y * v.y, b0y = cs.y * v.x;
    return mk2(fma_(cs.x, v.x, -b0x), fma_(cs.x, v.y, b0y));
}

// scalar Chebyshev of apply_pll_scalar (chebyshev_sine.h:22-41, no fused operations)
__device__ __forceinline__ float cheb_scalar(float x) {
    const float z = x * x;
    const float b4 = 3.20396066f * z + -14.07150173f;
    const float b3 = b4 * z + 38.50016403f;
    const float b2 = b3 * z + -67.07687378f;
    const float b1 = b2 * z + 64.83583069f;
    const float b0 = b1 * z + -25.13274193f;
    return b0 * (z - 0.25f) * x;
}

// sample n of a symbol of `period` samples: the vector body in groups of 4, then the scalar tail (apply_pll.cpp:12-30)
__device__ __forceinline__ f2 pll_any(f2 v, int n, int period, float f, float dt0) {
    const int nv = period & ~3;
    if (n < nv) {
        const int k = n & 3;
        const float ss = (float)k * f;
        return pll1(v, dt0 + (float)(n & ~3) * f, mk2(ss + 0.25f, ss));
    }
    const float dt_scalar = dt0 + (float)nv * f;
    float dt_sin = dt_scalar + (float)(n - nv) * f;
    float dt_cos = dt_sin + 0.25f;
    dt_sin = dt_sin - __builtin_roundf(dt_sin);
    dt_cos = dt_cos - __builtin_roundf(dt_cos);
    const float c = cheb_scalar(dt_cos), s = cheb_scalar(dt_sin);
    return mk2(v.x * c - v.y * s, v.x * s + v.y * c);
}

// x0 * conj(x1), FMA form of x86/c32_conj_mul.h:12-44
__device__ __forceinline__ f2 conj_mul(f2 x0, f2 x1) {
    const float a = x0.x, b = x0.y, c = x1.x, d = x1.y;
    return mk2(fma_(b, d, a * c), fma_(b, c, -(a * d)));
}

// ---- FFT butterflies (contract identical to oracle/dab_oracle_ofdm.c dft4/dft8/cmul) ----
__device__ __forceinline__ f2 mul_mi(f2 a) { return mk2(a.y, -a.x); }
__device__ __forceinline__ f2 cmul(f2 b, f2 w) {
    const float t0 = b.y * w.y, t1 = b.y * w.x;          // (b.im*w.im, b.im*w.re)
    return mk2(fma_(b.x, w.x, -t0), fma_(b.x, w.y, t1));
}
constexpr float SQRT_HALF = 0.707106769084930420f;
__device__ __forceinline__ f2 mul_w8_1(f2 a) { return mk2((a.x + a.y) * SQRT_HALF, (a.y - a.x) * SQRT_HALF); }
__device__ __forceinline__ f2 mul_w8_3(f2 a) { return mk2((a.y - a.x) * SQRT_HALF, -((a.x + a.y) * SQRT_HALF)); }

__device__ __forceinline__ void dft4(const f2 a0, const f2 a1, const f2 a2, const f2 a3, f2& b0, f2& b1, f2& b2, f2& b3) {
    const f2 s02 = a0 + a2, d02 = a0 - a2;
    const f2 s13 = a1 + a3, d13 = mul_mi(a1 - a3);
    b0 = s02 + s13; b1 = d02 + d13; b2 = s02 - s13; b3 = d02 - d13;
}

__device__ __forceinline__ void dft8(f2 (&a)[8]) {
    const f2 c0 = a[0] + a[4], c1 = a[0] - a[4];
    const f2 c2 = a[2] + a[6], c3 = mul_mi(a[2] - a[6]);
    const f2 c4 = a[1] + a[5], c5 = a[1] - a[5];
    const f2 c6 = a[3] + a[7], c7 = mul_mi(a[3] - a[7]);
    const f2 d0 = c0 + c2, d2 = c0 - c2;
    const f2 d1 = c1 + c3, d3 = c1 - c3;
    const f2 d4 = c4 + c6, d6 = mul_mi(c4 - c6);
    const f2 d5 = mul_w8_1(c5 + c7), d7 = mul_w8_3(c5 - c7);
    a[0] = d0 + d4; a[4] = d0 - d4;
    a[1] = d1 + d5; a[5] = d1 - d5;
    a[2] = d2 + d6; a[6] = d2 - d6;
    a[3] = d3 + d7; a[7] = d3 - d7;
}

// convert_to_viterbi_bit (ofdm_demodulator.cpp:57-72): (int8)(-x*127); v_cvt_i32_f32 truncates and maps NaN to 0
__device__ __forceinline__ int to_vbit(float x) {
    const float v = -x * 127.0f;
    return (int)v;
}

// soft bits of one DQPSK product d: to_vbit(d.x / A), to_vbit(-(d.y / A)), A = max(|d.x|, |d.y|) (ofdm_demodulator.cpp:867-889).
// Inside [2^-60, 2^60] the two correctly rounded quotients share the reciprocal refinement of the compiler's own expansion
// (ofdm_demod.hip explains why the result is bit-identical); outside, or for NaN, the plain divisions run.
__device__ __forceinline__ void soft_bit_pair(f2 d, int& bx, int& by) {
    const float ar = __builtin_fabsf(d.x), ai = __builtin_fabsf(d.y);
    const float A = (ar < ai) ? ai : ar;                         // std::max, ofdm_demodulator.cpp:882
    if (A >= 0x1p-60f && A <= 0x1p60f) {
        const float nx = d.x, ny = -d.y;
        float r = __builtin_amdgcn_rcpf(A);
        r = fma_(fma_(-A, r, 1.0f), r, r);
        float qx = nx * r, qy = ny * r;
        qx = fma_(fma_(-A, qx, nx), r, qx);
        qy = fma_(fma_(-A, qy, ny), r, qy);
        qx = fma_(fma_(-A, qx, nx), r, qx);
        qy = fma_(fma_(-A, qy, ny), r, qy);
        bx = to_vbit(qx); by = to_vbit(qy);
    } else {
        bx = to_vbit(d.x / A); by = to_vbit(-(d.y / A));
    }
}

// ---- deterministic atan2 (same operation sequence as the oracle's dab_atan2f) ----
__device__ __forceinline__ float atan2_det(float y, float x) {
    const float PI_F = 3.14159274101257324f, PIO2_F = 1.57079637050628662f, PIO4_F = 0.785398185253143311f;
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = (ax > ay) ? ax : ay;
    const float mn = (ax > ay) ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float base = 0.0f;
    if (a > 0.4142135679721832f) { base = PIO4_F; a = (a - 1.0f) / (a + 1.0f); }
    const float z = a * a;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fma_(p, z, 1.99777106478e-1f);
    p = fma_(p, z, -3.33329491539e-1f);
    float r = fma_(p * z, a, a);
    r = base + r;
    if (ay > ax) r = PIO2_F - r;
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

// fine-frequency loop of one frame from its summed cyclic-prefix phase (ofdm_demodulator.cpp:779-824, :829-840)
__device__ __forceinline__ float fine_freq_update(float fine, float total, float beta, int n_sym, int n_fft) {
    const float TWO_PI = 3.14159274101257324f * 2.0f;
    const float avg = total / (float)n_sym;
    const float spacing = 1.0f / (float)n_fft;
    const float err = spacing * avg / TWO_PI;
    const float delta = -beta * err;
    const float wrap = 0.5f * spacing * 1.01f;
    return fmodf(fine + delta, wrap);
}

// value of `v` held by lane (lane ^ XORMASK), XORMASK in {32,16,8,4,2,1}, without touching the LDS crossbar:
// v_permlane32_swap / v_permlane16_swap (gfx950) and DPP row_ror:8 / bank-masked row shifts / quad_perm
template <int XORMASK>
__device__ __forceinline__ float lane_xor(float v, int lane) {
    const int i = __float_as_int(v);
    int r;
    if constexpr (XORMASK == 32) {
        const auto t = __builtin_amdgcn_permlane32_swap((unsigned)i, (unsigned)i, false, false);
        r = (int)((lane & 32) ? t[0] : t[1]);
    } else if constexpr (XORMASK == 16) {
        const auto t = __builtin_amdgcn_permlane16_swap((unsigned)i, (unsigned)i, false, false);
        r = (int)((lane & 16) ? t[0] : t[1]);
    } else if constexpr (XORMASK == 8) {
        r = __builtin_amdgcn_mov_dpp(i, 0x128, 0xF, 0xF, true);                      // row_ror:8
    } else if constexpr (XORMASK == 4) {
        const int a = __builtin_amdgcn_update_dpp(i, i, 0x104, 0xF, 0x5, false);      // row_shl:4 -> banks 0,2
        r = __builtin_amdgcn_update_dpp(a, i, 0x114, 0xF, 0xA, false);                // row_shr:4 -> banks 1,3
    } else if constexpr (XORMASK == 2) {
        r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true);                        // quad_perm [2,3,0,1]
    } else {
        r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true);                        // quad_perm [1,0,3,2]
    }
    return __int_as_float(r);
}

// the contract's 64-leaf reduction: a[i] += a[i ^ h] for h = 32, 16, ..., 1 (every lane ends with the same sum)
__device__ __forceinline__ float wave_tree_sum(float v, int lane) {
    v += lane_xor<32>(v, lane);
    v += lane_xor<16>(v, lane);
    v += lane_xor<8>(v, lane);
    v += lane_xor<4>(v, lane);
    v += lane_xor<2>(v, lane);
    v += lane_xor<1>(v, lane);
    return v;
}

// The same tree for a PAIR of values when only one lane needs each sum: stride 32 of both values costs one
// v_permlane32_swap (lanes 0..31 then carry x, lanes 32..63 carry y), the swap of stride 16 feeds both of its halves
// to one add, strides 8..1 are single row-shift DPP adds.  Every addition has the operands of the butterfly form above,
// so lane 0 ends with wave_tree_sum(x) and lane 32 with wave_tree_sum(y), bit for bit, in ~40 instead of ~100 issue cycles.
__device__ __forceinline__ float wave_tree_sum_pair(float x, float y) {
    const auto t32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    float s = __uint_as_float(t32[0]) + __uint_as_float(t32[1]);          // x[l] + x[l+32] | y[l-32] + y[l]
    const auto t16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    s = __uint_as_float(t16[0]) + __uint_as_float(t16[1]);                // rows 0 and 2: s[l] + s[l+16]
    s = s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x108, 0xF, 0xF, true));   // row_shl:8
    s = s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x104, 0xF, 0xF, true));   // row_shl:4
    s = s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x102, 0xF, 0xF, true));   // row_shl:2
    s = s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x101, 0xF, 0xF, true));   // row_shl:1
    return s;
}

// wave-private LDS hand-off: order this wave's LDS writes before its later reads for the compiler; the LDS unit
// itself executes one wave's instructions in order, so no s_barrier is involved
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- transmission-mode geometry (src/ofdm/dab_ofdm_params_ref.cpp:11-60) ----
// ---- size-generic Stockham autosort passes through LDS (modes II-IV; same butterflies and twiddle rule as mode I) ----
template <int R>
__device__ __forceinline__ void butterfly(f2 (&a)[8]) {
    if constexpr (R == 8) {
        dft8(a);
    } else if constexpr (R == 4) {
        f2 b0, b1, b2, b3;
        dft4(a[0], a[1], a[2], a[3], b0, b1, b2, b3);
        a[0] = b0; a[1] = b1; a[2] = b2; a[3] = b3;
    } else {
        const f2 s = a[0] + a[1], d = a[0] - a[1];
        a[0] = s; a[1] = d;
    }
}

// one Stockham pass of radix R over `src` (current sub-transform length cur_n, stride s) into `dst` -- both in LDS: the callers ping-pong
// between two LDS arrays chosen at run time, which leaves the compiler with generic pointers (flat_load / flat_store, slower than the
// LDS path and counted on both memory counters); the casts say what they are
typedef float lds_v2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lds_v2* lds_cf2;
typedef __attribute__((address_space(3))) lds_v2* lds_f2;
__device__ __forceinline__ f2 lds_get(lds_cf2 p, int i) { const lds_v2 v = p[i]; return mk2(v.x, v.y); }
__device__ __forceinline__ void lds_put(lds_f2 p, int i, f2 v) { lds_v2 w; w.x = v.x; w.y = v.y; p[i] = w; }
template <int R>
__device__ __forceinline__ void stockham_pass(const f2* src_, f2* dst_, int n_total, int cur_n, int s, bool last,
                                              const f2* __restrict__ tw, int t, int n_threads) {
    const lds_cf2 src = (lds_cf2)src_;
    const lds_f2 dst = (lds_f2)dst_;
    const int m = cur_n / R, tw_step = NB_FFT / cur_n;
    for (int u = t; u < n_total / R; u += n_threads) {
        const int q = u % s, p = u / s;
        f2 a[8];
#pragma unroll
        for (int j = 0; j < R; j++) a[j] = lds_get(src, q + s * (p + m * j));
        butterfly<R>(a);
#pragma unroll
        for (int k = 0; k < R; k++) lds_put(dst, q + s * (R * p + k), (k == 0 || last) ? a[k] : cmul(a[k], tw[tw_step * p * k]));
    }
}

}  // namespace dabgpu
