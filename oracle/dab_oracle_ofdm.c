/*
 * dab_oracle_ofdm.c -- CPU ORACLE (test infrastructure, NOT product code): OFDM half.
 * See dab_oracle.h for the scope statement and the arithmetic contract.
 * Paths cited are relative to /root/reference.
 */
#include "dab_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
/* same arithmetic, two code paths: hardware vfmadd when the CPU has it, libm fmaf otherwise */
#define DAB_HOT __attribute__((target_clones("arch=haswell", "default")))
#else
#define DAB_HOT
#endif

/* ------------------------------------------------------------------------------------------ */
/* constant tables                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* src/ofdm/dab_mapper_ref.cpp:10-51 */
void dab_get_mapper(int *carrier_map) {
    const int N = DAB_NB_FFT, K = N / 4, nb = DAB_NB_DATA_CARRIERS;
    const int dc = N / 2, lo = dc - nb / 2, hi = dc + nb / 2;
    int pi = 0, k = 0;
    for (int i = 0; i < N; i++) {
        if (i > 0) pi = (13 * pi + K - 1) % N;              /* :24 */
        if (pi < lo || pi > hi || pi == dc) continue;       /* :39 */
        carrier_map[k++] = (pi < dc) ? (pi - lo) : (pi - lo - 1);  /* :43-49 */
    }
}

/* ETSI EN 300 401 table 23/24 as transcribed in src/ofdm/dab_prs_ref.cpp:24-75 (Mode I rows) and :125-130 */
static const signed char PRS_I_IDX[48] = {
    0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3,
    0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1 };
static const signed char PRS_I_N[48] = {
    1,2,0,1, 3,2,2,3, 2,1,2,3, 1,2,3,3, 2,2,2,1, 1,3,1,2,
    3,1,1,1, 2,2,1,0, 2,2,3,3, 0,2,1,3, 3,3,3,0, 3,0,1,1 };
static const signed char PRS_H[4][32] = {
    {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1,0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1},
    {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0,0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
    {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3,0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3},
    {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2,0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2},
};

/* src/ofdm/dab_prs_ref.cpp:140-195 */
void dab_get_prs_fft(dab_cf32 *prs) {
    for (int i = 0; i < DAB_NB_FFT; i++) { prs[i].re = 0.0f; prs[i].im = 0.0f; }
    for (int row = 0; row < 48; row++) {
        /* rows 0..23 cover k = -768..-1, rows 24..47 cover k = +1..+768, 32 carriers each */
        const int k_min = (row < 24) ? (-768 + 32 * row) : (1 + 32 * (row - 24));
        for (int j = 0; j < 32; j++) {
            const int k = k_min + j;
            const int h = PRS_H[(int)PRS_I_IDX[row]][j];
            const float phi = (float)M_PI / 2.0f * (float)(h + PRS_I_N[row]);   /* :167,:184 */
            dab_cf32 v; v.re = cosf(phi); v.im = sinf(phi);
            prs[(k < 0) ? (DAB_NB_FFT + k) : k] = v;
        }
    }
}

void dab_get_twiddles(dab_cf32 *tw) {
    for (int m = 0; m < DAB_NB_FFT; m++) {
        const double a = 2.0 * M_PI * (double)m / (double)DAB_NB_FFT;
        tw[m].re = (float)cos(a);
        tw[m].im = (float)(-sin(a));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DSP primitives                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* src/ofdm/dsp/chebyshev_sine.h:13-20 */
static const float CH_A0 = -25.13274193f, CH_A1 = 64.83583069f, CH_A2 = -67.07687378f,
                   CH_A3 = 38.50016403f,  CH_A4 = -14.07150173f, CH_A5 = 3.20396066f;

/* src/ofdm/dsp/chebyshev_sine.h:22-41 */
float dab_chebyshev_sine(float x) {
    const float z = x * x;
    const float b4 = CH_A5 * z + CH_A4;
    const float b3 = b4 * z + CH_A3;
    const float b2 = b3 * z + CH_A2;
    const float b1 = b2 * z + CH_A1;
    const float b0 = b1 * z + CH_A0;
    return b0 * (z - 0.25f) * x;
}

/* src/ofdm/dsp/chebyshev_sine.h:82-107 with __FMA__ */
static inline float cheb_fma(float x) {
    const float z = x * x;
    const float b4 = fmaf(CH_A5, z, CH_A4);
    const float b3 = fmaf(b4, z, CH_A3);
    const float b2 = fmaf(b3, z, CH_A2);
    const float b1 = fmaf(b2, z, CH_A1);
    const float b0 = fmaf(b1, z, CH_A0);
    const float c0 = z - 0.25f;
    return (b0 * c0) * x;
}
float dab_chebyshev_sine_fma(float x) { return cheb_fma(x); }

/* src/ofdm/dsp/apply_pll.cpp:12-30 ; std::complex product without contraction */
void dab_apply_pll_scalar(const dab_cf32 *x, dab_cf32 *y, size_t n, float freq_norm, float dt_norm) {
    for (size_t i = 0; i < n; i++) {
        float dt_sin = dt_norm + (float)i * freq_norm;
        float dt_cos = dt_sin + 0.25f;
        dt_sin = dt_sin - roundf(dt_sin);
        dt_cos = dt_cos - roundf(dt_cos);
        const float c = dab_chebyshev_sine(dt_cos);
        const float s = dab_chebyshev_sine(dt_sin);
        const float xr = x[i].re, xi = x[i].im;
        y[i].re = xr * c - xi * s;
        y[i].im = xr * s + xi * c;
    }
}

/* one sample of apply_pll_avx (apply_pll.cpp:81-117) + c32_mul_avx with __FMA__ (x86/c32_mul.h:9-38):
 * i4 = index of the 4-sample group start, k = lane in the group */
static inline dab_cf32 pll_sample(dab_cf32 v, size_t i4, int k, float f, float dt_norm) {
    const float step_sin = (float)k * f;                 /* :95-99 */
    const float step_cos = step_sin + 0.25f;
    const float base = dt_norm + (float)i4 * f;          /* :103 */
    float dc = base + step_cos;                          /* :104 */
    float ds = base + step_sin;
    dc = dc - rintf(dc);                                 /* :107 round-to-nearest-even */
    ds = ds - rintf(ds);
    const float c = cheb_fma(dc), s = cheb_fma(ds);
    /* c32_mul_avx(X, pll): b0 = [s*xi, s*xr]; y = fmaddsub([c,c],[xr,xi],b0) */
    const float b0r = s * v.im, b0i = s * v.re;
    dab_cf32 y;
    y.re = fmaf(c, v.re, -b0r);
    y.im = fmaf(c, v.im, b0i);
    return y;
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
#include <immintrin.h>
/* The same element arithmetic eight samples at a time (two groups of four): every operation below is the lane-wise form of one line of
 * pll_sample -- same operands, same order, same roundings (vfmadd = fmaf, vroundps nearest-even = rintf) -- so the result is pll_sample's
 * bit for bit (tests/test_oracle_pins.py holds both to the reference's AVX2 object).  Without it the CPU baseline of bench.py would time a
 * scalar emulation of a loop the reference runs in AVX2 (tools/cpu_calibration.py). */
__attribute__((target("avx2,fma")))
static size_t pll_groups_avx2(const dab_cf32 *x, dab_cf32 *y, size_t nv, float f, float dt_norm) {
    const __m256 kf = _mm256_setr_ps(0.0f, 1.0f, 2.0f, 3.0f, 0.0f, 1.0f, 2.0f, 3.0f);
    const __m256 step_sin = _mm256_mul_ps(kf, _mm256_set1_ps(f));
    const __m256 step_cos = _mm256_add_ps(step_sin, _mm256_set1_ps(0.25f));
    const __m256 a5 = _mm256_set1_ps(CH_A5), a4 = _mm256_set1_ps(CH_A4), a3 = _mm256_set1_ps(CH_A3), a2 = _mm256_set1_ps(CH_A2),
                 a1 = _mm256_set1_ps(CH_A1), a0 = _mm256_set1_ps(CH_A0), q = _mm256_set1_ps(0.25f);
    const __m256i deint = _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7);
    size_t i = 0;
    for (; i + 8 <= nv; i += 8) {
        const float b0 = dt_norm + (float)i * f, b1 = dt_norm + (float)(i + 4) * f;
        const __m256 base = _mm256_setr_ps(b0, b0, b0, b0, b1, b1, b1, b1);
        __m256 dc = _mm256_add_ps(base, step_cos), ds = _mm256_add_ps(base, step_sin);
        dc = _mm256_sub_ps(dc, _mm256_round_ps(dc, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
        ds = _mm256_sub_ps(ds, _mm256_round_ps(ds, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
        __m256 zc = _mm256_mul_ps(dc, dc), zs = _mm256_mul_ps(ds, ds);
        __m256 pc = _mm256_fmadd_ps(a5, zc, a4), ps = _mm256_fmadd_ps(a5, zs, a4);
        pc = _mm256_fmadd_ps(pc, zc, a3); ps = _mm256_fmadd_ps(ps, zs, a3);
        pc = _mm256_fmadd_ps(pc, zc, a2); ps = _mm256_fmadd_ps(ps, zs, a2);
        pc = _mm256_fmadd_ps(pc, zc, a1); ps = _mm256_fmadd_ps(ps, zs, a1);
        pc = _mm256_fmadd_ps(pc, zc, a0); ps = _mm256_fmadd_ps(ps, zs, a0);
        const __m256 c = _mm256_mul_ps(_mm256_mul_ps(pc, _mm256_sub_ps(zc, q)), dc);
        const __m256 s = _mm256_mul_ps(_mm256_mul_ps(ps, _mm256_sub_ps(zs, q)), ds);
        /* eight interleaved (re, im) pairs -> re[8], im[8] in sample order */
        const __m256 v0 = _mm256_loadu_ps((const float *)(x + i)), v1 = _mm256_loadu_ps((const float *)(x + i + 4));
        const __m256 p0 = _mm256_permutevar8x32_ps(v0, deint), p1 = _mm256_permutevar8x32_ps(v1, deint);     /* re0..3 im0..3 | re4..7 im4..7 */
        const __m256 re = _mm256_permute2f128_ps(p0, p1, 0x20), im = _mm256_permute2f128_ps(p0, p1, 0x31);
        const __m256 yr = _mm256_fmadd_ps(c, re, _mm256_xor_ps(_mm256_mul_ps(s, im), _mm256_set1_ps(-0.0f)));   /* fmaf(c, re, -(s * im)) */
        const __m256 yi = _mm256_fmadd_ps(c, im, _mm256_mul_ps(s, re));                                        /* fmaf(c, im, s * re) */
        const __m256 lo = _mm256_unpacklo_ps(yr, yi), hi = _mm256_unpackhi_ps(yr, yi);                         /* per 128-bit half: pairs 0,1 | 2,3 */
        _mm256_storeu_ps((float *)(y + i), _mm256_permute2f128_ps(lo, hi, 0x20));
        _mm256_storeu_ps((float *)(y + i + 4), _mm256_permute2f128_ps(lo, hi, 0x31));
    }
    return i;
}
#endif

DAB_HOT
void dab_apply_pll(const dab_cf32 *x, dab_cf32 *y, size_t n, float freq_norm, float dt_norm) {
    const size_t nv = (n / 4) * 4;
    size_t i = 0;
#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) i = pll_groups_avx2(x, y, nv, freq_norm, dt_norm);
#endif
    for (; i < nv; i += 4)
        for (int k = 0; k < 4; k++)
            y[i + k] = pll_sample(x[i + k], i, k, freq_norm, dt_norm);
    if (nv < n) {                                        /* apply_pll.cpp:115-116 */
        const float dt_scalar = dt_norm + (float)nv * freq_norm;
        dab_apply_pll_scalar(x + nv, y + nv, n - nv, freq_norm, dt_scalar);
    }
}

/* element of c32_conj_mul_avx with __FMA__ (x86/c32_conj_mul.h:12-44): x0*conj(x1) */
static inline dab_cf32 conj_mul(dab_cf32 x0, dab_cf32 x1) {
    const float a = x0.re, b = x0.im, c = x1.re, d = x1.im;
    dab_cf32 y;
    y.re = fmaf(b, d, a * c);        /* bd + ac */
    y.im = fmaf(b, c, -(a * d));     /* bc - ad */
    return y;
}
dab_cf32 dab_conj_mul(dab_cf32 x0, dab_cf32 x1) { return conj_mul(x0, x1); }

/* fixed reduction tree used for every 256-leaf sum in the contract:
 * per 64-leaf group halve with strides 32..1, then (g0+g1)+(g2+g3). */
static inline float tree256(float *a) {
    for (int w = 0; w < 4; w++)
        for (int h = 32; h >= 1; h >>= 1)
            for (int i = 0; i < h; i++) a[64 * w + i] += a[64 * w + i + h];
    return (a[0] + a[64]) + (a[128] + a[192]);
}

/* ofdm_demodulator.cpp:768-775 */
DAB_HOT
dab_cf32 dab_cp_correlation(const dab_cf32 *sym) {
    float pr[256], pi[256];
    for (int t = 0; t < 256; t++) {
        if (t < 4) { pr[t] = 0.0f; pi[t] = 0.0f; continue; }
        const int n = 2 * (t - 4);
        const dab_cf32 p0 = conj_mul(sym[DAB_NB_FFT + n],     sym[n]);
        const dab_cf32 p1 = conj_mul(sym[DAB_NB_FFT + n + 1], sym[n + 1]);
        pr[t] = p0.re + p1.re;
        pi[t] = p0.im + p1.im;
    }
    dab_cf32 r; r.re = tree256(pr); r.im = tree256(pi);
    return r;
}

/* deterministic atan2 (range reduction + odd minimax polynomial, Cephes atanf constants) */
float dab_atan2f(float y, float x) {
    const float PI_F = 3.14159274101257324f, PIO2_F = 1.57079637050628662f, PIO4_F = 0.785398185253143311f;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = (ax > ay) ? ax : ay;
    const float mn = (ax > ay) ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float base = 0.0f;
    if (a > 0.4142135679721832f) { base = PIO4_F; a = (a - 1.0f) / (a + 1.0f); }
    const float z = a * a;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = fmaf(p * z, a, a);
    r = base + r;
    if (ay > ax) r = PIO2_F - r;
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

float dab_cabsf(dab_cf32 v) { return sqrtf(fmaf(v.re, v.re, v.im * v.im)); }

/* 20*log10(m): exponent split + Cephes logf polynomial on [sqrt(.5), sqrt(2)) */
float dab_db20f(float m) {
    if (!(m > 0.0f)) return -INFINITY;
    union { float f; uint32_t u; } b;
    float e_adj = 0.0f;
    if (m < 1.17549435e-38f) { m = m * 16777216.0f; e_adj = -24.0f; }
    b.f = m;
    int e = (int)((b.u >> 23) & 0xFF) - 126;                 /* m = f * 2^e, f in [0.5,1) */
    b.u = (b.u & 0x007FFFFFu) | 0x3F000000u;
    float f = b.f;
    if (f < 0.707106769084930420f) { e -= 1; f = f + f; }    /* f in [sqrt(.5), sqrt(2)) */
    const float x = f - 1.0f;
    const float z = x * x;
    float p = fmaf(7.0376836292e-2f, x, -1.1514610310e-1f);
    p = fmaf(p, x, 1.1676998740e-1f);
    p = fmaf(p, x, -1.2420140846e-1f);
    p = fmaf(p, x, 1.4249322787e-1f);
    p = fmaf(p, x, -1.6668057665e-1f);
    p = fmaf(p, x, 2.0000714765e-1f);
    p = fmaf(p, x, -2.4999993993e-1f);
    p = fmaf(p, x, 3.3333331174e-1f);
    float y = (p * x) * z;
    y = fmaf(-0.5f, z, y);
    const float fe = (float)e + e_adj;
    const float ln = fmaf(fe, 0.693147182464599609f, x + y);
    return ln * 8.68588924407958984f;                        /* 20/ln(10) */
}

/* 10^(db/20) = 2^(db*log2(10)/20), Cephes exp2f polynomial */
float dab_undb20f(float db) {
    float x = db * 0.166096404194831848f;
    if (x > 127.0f) return INFINITY;
    if (!(x > -149.0f)) return 0.0f;
    float n = floorf(x);
    x = x - n;
    if (x > 0.5f) { n += 1.0f; x -= 1.0f; }
    float p = fmaf(1.535336188319500e-4f, x, 1.339887440266574e-3f);
    p = fmaf(p, x, 9.618437357674640e-3f);
    p = fmaf(p, x, 5.550332471162809e-2f);
    p = fmaf(p, x, 2.402264791363012e-1f);
    p = fmaf(p, x, 6.931472028550421e-1f);
    p = fmaf(p, x, 1.0f);
    return ldexpf(p, (int)n);
}

/* ------------------------------------------------------------------------------------------ */
/* FFT contract: Stockham autosort, 2048 = 4 x 8 x 8 x 8, decimation in frequency               */
/* ------------------------------------------------------------------------------------------ */

static dab_cf32 g_tw[DAB_NB_FFT];
static int g_tw_ready = 0;
static void ensure_tw1(void);
static void ensure_tw(void) { if (!g_tw_ready) { dab_get_twiddles(g_tw); ensure_tw1(); g_tw_ready = 1; } }

static inline dab_cf32 cadd(dab_cf32 a, dab_cf32 b) { dab_cf32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline dab_cf32 csub(dab_cf32 a, dab_cf32 b) { dab_cf32 r = { a.re - b.re, a.im - b.im }; return r; }
static inline dab_cf32 mul_mi(dab_cf32 a) { dab_cf32 r = { a.im, -a.re }; return r; }       /* * (-i) */
/* b*w : re = fma(b.re, w.re, -(b.im*w.im)), im = fma(b.re, w.im, b.im*w.re) */
static inline dab_cf32 cmul(dab_cf32 b, dab_cf32 w) {
    const float t0 = b.im * w.im, t1 = b.im * w.re;
    dab_cf32 r; r.re = fmaf(b.re, w.re, -t0); r.im = fmaf(b.re, w.im, t1);
    return r;
}
#define SQRT_HALF 0.707106769084930420f
/* * (1-i)/sqrt2 and * (-1-i)/sqrt2 */
static inline dab_cf32 mul_w8_1(dab_cf32 a) { dab_cf32 r = { (a.re + a.im) * SQRT_HALF, (a.im - a.re) * SQRT_HALF }; return r; }
static inline dab_cf32 mul_w8_3(dab_cf32 a) { dab_cf32 r = { (a.im - a.re) * SQRT_HALF, -((a.re + a.im) * SQRT_HALF) }; return r; }

static inline void dft4(const dab_cf32 *a, dab_cf32 *b) {
    const dab_cf32 s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
    const dab_cf32 s13 = cadd(a[1], a[3]), d13 = mul_mi(csub(a[1], a[3]));
    b[0] = cadd(s02, s13); b[1] = cadd(d02, d13);
    b[2] = csub(s02, s13); b[3] = csub(d02, d13);
}

static inline void dft8(const dab_cf32 *a, dab_cf32 *b) {
    const dab_cf32 c0 = cadd(a[0], a[4]), c1 = csub(a[0], a[4]);
    const dab_cf32 c2 = cadd(a[2], a[6]), c3 = mul_mi(csub(a[2], a[6]));
    const dab_cf32 c4 = cadd(a[1], a[5]), c5 = csub(a[1], a[5]);
    const dab_cf32 c6 = cadd(a[3], a[7]), c7 = mul_mi(csub(a[3], a[7]));
    const dab_cf32 d0 = cadd(c0, c2), d2 = csub(c0, c2);
    const dab_cf32 d1 = cadd(c1, c3), d3 = csub(c1, c3);
    const dab_cf32 d4 = cadd(c4, c6), d6 = mul_mi(csub(c4, c6));
    const dab_cf32 d5 = mul_w8_1(cadd(c5, c7)), d7 = mul_w8_3(csub(c5, c7));
    b[0] = cadd(d0, d4); b[4] = csub(d0, d4);
    b[1] = cadd(d1, d5); b[5] = csub(d1, d5);
    b[2] = cadd(d2, d6); b[6] = csub(d2, d6);
    b[3] = cadd(d3, d7); b[7] = csub(d3, d7);
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
/* The same transform four butterflies at a time: a vector holds four complex values of the pass's independent index (interleaved re, im),
 * and every vector operation below is the lane-wise form of one scalar helper above -- cadd / csub (add, sub), mul_mi (swap + sign),
 * mul_w8_1 / mul_w8_3 (the same sum / difference, the same product with SQRT_HALF), cmul (t = b.im * (w.im, w.re); re = fma(b.re, w.re, -t0),
 * im = fma(b.re, w.im, t1)) -- same operands, same order, same roundings: bit-identical outputs (tests/test_oracle_pins.py and
 * test_oracle_properties.py compare the two paths through DAB_ORACLE_SCALAR_FFT).  Only the CPU baseline's speed changes. */
#define V8 __m256
static dab_cf32 g_tw1[3][512];                       /* pass-1 twiddles g_tw[p * k], contiguous in p */
__attribute__((target("avx2,fma"))) static inline V8 v_mul_mi(V8 a) {                        /* (im, -re) */
    const V8 odd = _mm256_castsi256_ps(_mm256_setr_epi32(0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u));
    return _mm256_xor_ps(_mm256_permute_ps(a, 0xB1), odd);
}
__attribute__((target("avx2,fma"))) static inline V8 v_cmul(V8 b, V8 w) {
    const V8 even = _mm256_castsi256_ps(_mm256_setr_epi32((int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u, 0));
    const V8 t = _mm256_xor_ps(_mm256_mul_ps(_mm256_movehdup_ps(b), _mm256_permute_ps(w, 0xB1)), even);      /* (-(b.im w.im), b.im w.re) */
    return _mm256_fmadd_ps(_mm256_moveldup_ps(b), w, t);
}
__attribute__((target("avx2,fma"))) static inline V8 v_mul_w8_1(V8 a) {                      /* ((re + im) S, (im - re) S) */
    const V8 sw = _mm256_permute_ps(a, 0xB1);
    return _mm256_mul_ps(_mm256_blend_ps(_mm256_add_ps(a, sw), _mm256_sub_ps(a, sw), 0xAA), _mm256_set1_ps(SQRT_HALF));
}
__attribute__((target("avx2,fma"))) static inline V8 v_mul_w8_3(V8 a) {                      /* ((im - re) S, -((re + im) S)) */
    const V8 odd = _mm256_castsi256_ps(_mm256_setr_epi32(0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u));
    const V8 sw = _mm256_permute_ps(a, 0xB1);
    return _mm256_xor_ps(_mm256_mul_ps(_mm256_blend_ps(_mm256_sub_ps(sw, a), _mm256_add_ps(a, sw), 0xAA), _mm256_set1_ps(SQRT_HALF)), odd);
}
__attribute__((target("avx2,fma"))) static inline void v_dft8(const V8 *a, V8 *b) {
    const V8 c0 = _mm256_add_ps(a[0], a[4]), c1 = _mm256_sub_ps(a[0], a[4]);
    const V8 c2 = _mm256_add_ps(a[2], a[6]), c3 = v_mul_mi(_mm256_sub_ps(a[2], a[6]));
    const V8 c4 = _mm256_add_ps(a[1], a[5]), c5 = _mm256_sub_ps(a[1], a[5]);
    const V8 c6 = _mm256_add_ps(a[3], a[7]), c7 = v_mul_mi(_mm256_sub_ps(a[3], a[7]));
    const V8 d0 = _mm256_add_ps(c0, c2), d2 = _mm256_sub_ps(c0, c2);
    const V8 d1 = _mm256_add_ps(c1, c3), d3 = _mm256_sub_ps(c1, c3);
    const V8 d4 = _mm256_add_ps(c4, c6), d6 = v_mul_mi(_mm256_sub_ps(c4, c6));
    const V8 d5 = v_mul_w8_1(_mm256_add_ps(c5, c7)), d7 = v_mul_w8_3(_mm256_sub_ps(c5, c7));
    b[0] = _mm256_add_ps(d0, d4); b[4] = _mm256_sub_ps(d0, d4);
    b[1] = _mm256_add_ps(d1, d5); b[5] = _mm256_sub_ps(d1, d5);
    b[2] = _mm256_add_ps(d2, d6); b[6] = _mm256_sub_ps(d2, d6);
    b[3] = _mm256_add_ps(d3, d7); b[7] = _mm256_sub_ps(d3, d7);
}
#define VLD(p) _mm256_loadu_ps((const float *)(p))
#define VST(p, v) _mm256_storeu_ps((float *)(p), (v))
#define VBC(w) _mm256_castpd_ps(_mm256_broadcast_sd((const double *)&(w)))              /* one complex value in all four slots */
__attribute__((target("avx2,fma")))
static void fft2048_core_avx2(const dab_cf32 *in, dab_cf32 *out, int conj_io) {
    dab_cf32 x[DAB_NB_FFT], y[DAB_NB_FFT];
    const V8 cj = conj_io ? _mm256_castsi256_ps(_mm256_setr_epi32(0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u, 0, (int)0x80000000u))
                          : _mm256_setzero_ps();
    V8 a[8], b[8];
    /* pass 1: radix 4, four consecutive p per vector; the outputs of one p are contiguous: 4 x 4 transpose of complex values */
    for (int p = 0; p < 512; p += 4) {
        for (int j = 0; j < 4; j++) a[j] = _mm256_xor_ps(VLD(in + p + 512 * j), cj);
        const V8 s02 = _mm256_add_ps(a[0], a[2]), d02 = _mm256_sub_ps(a[0], a[2]);
        const V8 s13 = _mm256_add_ps(a[1], a[3]), d13 = v_mul_mi(_mm256_sub_ps(a[1], a[3]));
        b[0] = _mm256_add_ps(s02, s13);
        b[1] = v_cmul(_mm256_add_ps(d02, d13), VLD(&g_tw1[0][p]));
        b[2] = v_cmul(_mm256_sub_ps(s02, s13), VLD(&g_tw1[1][p]));
        b[3] = v_cmul(_mm256_sub_ps(d02, d13), VLD(&g_tw1[2][p]));
        const __m256d t0 = _mm256_unpacklo_pd(_mm256_castps_pd(b[0]), _mm256_castps_pd(b[1])), t1 = _mm256_unpackhi_pd(_mm256_castps_pd(b[0]), _mm256_castps_pd(b[1]));
        const __m256d t2 = _mm256_unpacklo_pd(_mm256_castps_pd(b[2]), _mm256_castps_pd(b[3])), t3 = _mm256_unpackhi_pd(_mm256_castps_pd(b[2]), _mm256_castps_pd(b[3]));
        VST(y + 4 * p,      _mm256_castpd_ps(_mm256_permute2f128_pd(t0, t2, 0x20)));
        VST(y + 4 * p + 4,  _mm256_castpd_ps(_mm256_permute2f128_pd(t1, t3, 0x20)));
        VST(y + 4 * p + 8,  _mm256_castpd_ps(_mm256_permute2f128_pd(t0, t2, 0x31)));
        VST(y + 4 * p + 12, _mm256_castpd_ps(_mm256_permute2f128_pd(t1, t3, 0x31)));
    }
    /* pass 2: radix 8, n = 512, s = 4: q = 0 .. 3 is one vector */
    for (int p = 0; p < 64; p++) {
        for (int j = 0; j < 8; j++) a[j] = VLD(y + 4 * (p + 64 * j));
        v_dft8(a, b);
        VST(x + 4 * (8 * p), b[0]);
        for (int k = 1; k < 8; k++) VST(x + 4 * (8 * p + k), v_cmul(b[k], VBC(g_tw[4 * p * k])));
    }
    /* pass 3: radix 8, n = 64, s = 32 */
    for (int p = 0; p < 8; p++) {
        V8 w[8];
        for (int k = 1; k < 8; k++) w[k] = VBC(g_tw[32 * p * k]);
        for (int q = 0; q < 32; q += 4) {
            for (int j = 0; j < 8; j++) a[j] = VLD(x + q + 32 * (p + 8 * j));
            v_dft8(a, b);
            VST(y + q + 32 * (8 * p), b[0]);
            for (int k = 1; k < 8; k++) VST(y + q + 32 * (8 * p + k), v_cmul(b[k], w[k]));
        }
    }
    /* pass 4: radix 8, n = 8, s = 256 */
    for (int q = 0; q < 256; q += 4) {
        for (int j = 0; j < 8; j++) a[j] = VLD(y + q + 256 * j);
        v_dft8(a, b);
        for (int k = 0; k < 8; k++) VST(out + q + 256 * k, _mm256_xor_ps(b[k], cj));
    }
}
#undef VLD
#undef VST
#undef VBC
#undef V8
static void ensure_tw1(void) { for (int k = 1; k < 4; k++) for (int p = 0; p < 512; p++) g_tw1[k - 1][p] = g_tw[p * k]; }
static int g_scalar_fft = -1;                        /* DAB_ORACLE_SCALAR_FFT=1: the scalar statement of the transform (the tests run both) */
#endif

#if !(defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES))
static void ensure_tw1(void) {}
#endif

DAB_HOT
static void fft2048_core(const dab_cf32 *in, dab_cf32 *out, int conj_io) {
#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
    if (g_scalar_fft < 0) { const char *e = getenv("DAB_ORACLE_SCALAR_FFT"); g_scalar_fft = (e && e[0] == '1') || !(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")); }
    if (!g_scalar_fft) { fft2048_core_avx2(in, out, conj_io); return; }
#endif
    dab_cf32 x[DAB_NB_FFT], y[DAB_NB_FFT];
    dab_cf32 a[8], b[8];
    /* pass 1: radix 4, n=2048, s=1 */
    for (int p = 0; p < 512; p++) {
        for (int j = 0; j < 4; j++) {
            a[j] = in[p + 512 * j];
            if (conj_io) a[j].im = -a[j].im;
        }
        dft4(a, b);
        y[4 * p] = b[0];
        for (int k = 1; k < 4; k++) y[4 * p + k] = cmul(b[k], g_tw[p * k]);
    }
    /* pass 2: radix 8, n=512, s=4 */
    for (int p = 0; p < 64; p++)
        for (int q = 0; q < 4; q++) {
            for (int j = 0; j < 8; j++) a[j] = y[q + 4 * (p + 64 * j)];
            dft8(a, b);
            x[q + 4 * (8 * p)] = b[0];
            for (int k = 1; k < 8; k++) x[q + 4 * (8 * p + k)] = cmul(b[k], g_tw[4 * p * k]);
        }
    /* pass 3: radix 8, n=64, s=32 */
    for (int p = 0; p < 8; p++)
        for (int q = 0; q < 32; q++) {
            for (int j = 0; j < 8; j++) a[j] = x[q + 32 * (p + 8 * j)];
            dft8(a, b);
            y[q + 32 * (8 * p)] = b[0];
            for (int k = 1; k < 8; k++) y[q + 32 * (8 * p + k)] = cmul(b[k], g_tw[32 * p * k]);
        }
    /* pass 4: radix 8, n=8, s=256 */
    for (int q = 0; q < 256; q++) {
        for (int j = 0; j < 8; j++) a[j] = y[q + 256 * j];
        dft8(a, b);
        for (int k = 0; k < 8; k++) {
            dab_cf32 v = b[k];
            if (conj_io) v.im = -v.im;
            out[q + 256 * k] = v;
        }
    }
}

void dab_fft2048(const dab_cf32 *in, dab_cf32 *out)  { ensure_tw(); fft2048_core(in, out, 0); }
/* IFFT(x) = conj(FFT(conj(x))), unnormalised like FFTW_BACKWARD */
void dab_ifft2048(const dab_cf32 *in, dab_cf32 *out) { ensure_tw(); fft2048_core(in, out, 1); }

void dab_dft_naive(const dab_cf32 *in, double *out_re, double *out_im, int n, int inverse) {
    const double sgn = inverse ? 1.0 : -1.0;
    for (int k = 0; k < n; k++) {
        double sr = 0.0, si = 0.0;
        for (int j = 0; j < n; j++) {
            const double a = sgn * 2.0 * M_PI * (double)(((long)j * k) % n) / (double)n;
            const double c = cos(a), s = sin(a);
            sr += in[j].re * c - in[j].im * s;
            si += in[j].re * s + in[j].im * c;
        }
        out_re[k] = sr; out_im[k] = si;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* per-frame demodulation                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* convert_to_viterbi_bit ofdm_demodulator.cpp:57-72 ; float->int8 truncation, NaN -> 0 (x86 cvttss2si low byte) */
static inline int8_t to_vbit(float x) {
    const float v = -x * 127.0f;
    if (v != v) return 0;
    return (int8_t)(int)v;
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
/* eight carriers at a time, lane by lane the statements of the scalar loop below: conj_mul's two FMAs, fabsf, the max as (ar < ai) ? ai : ar,
 * IEEE division, -x * 127, truncation (vcvttps2dq: NaN -> INT_MIN, whose low byte is the scalar code's 0) */
__attribute__((target("avx2,fma")))
static int demap_avx2(const dab_cf32 *fft_i, const dab_cf32 *fft_ip1, const int *mapper, int8_t *bits, int n_vec) {
    const int N = DAB_NB_DATA_CARRIERS, M = N / 2;
    const __m256i deint = _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7);
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF)), sign = _mm256_set1_ps(-0.0f), k127 = _mm256_set1_ps(127.0f);
    const __m128i pick = _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    int i = 0;
    for (; i + 8 <= n_vec; i += 8) {
        int bin[8];
        for (int k = 0; k < 8; k++) {
            const int c = mapper[i + k];
            const int kk = (c < M) ? (c - M) : (c - M + 1);
            bin[k] = (DAB_NB_FFT + kk) % DAB_NB_FFT;
        }
        const __m128i b_lo = _mm_loadu_si128((const __m128i *)bin), b_hi = _mm_loadu_si128((const __m128i *)(bin + 4));
        /* gathers of 64-bit (re, im) pairs, then re[8] / im[8] in carrier order */
        const __m256 x0a = _mm256_castpd_ps(_mm256_i32gather_pd((const double *)fft_i, b_lo, 8)), x0b = _mm256_castpd_ps(_mm256_i32gather_pd((const double *)fft_i, b_hi, 8));
        const __m256 x1a = _mm256_castpd_ps(_mm256_i32gather_pd((const double *)fft_ip1, b_lo, 8)), x1b = _mm256_castpd_ps(_mm256_i32gather_pd((const double *)fft_ip1, b_hi, 8));
        const __m256 p0a = _mm256_permutevar8x32_ps(x0a, deint), p0b = _mm256_permutevar8x32_ps(x0b, deint);
        const __m256 p1a = _mm256_permutevar8x32_ps(x1a, deint), p1b = _mm256_permutevar8x32_ps(x1b, deint);
        const __m256 a = _mm256_permute2f128_ps(p0a, p0b, 0x20), b = _mm256_permute2f128_ps(p0a, p0b, 0x31);       /* x0 = fft_i:   re, im */
        const __m256 c = _mm256_permute2f128_ps(p1a, p1b, 0x20), d = _mm256_permute2f128_ps(p1a, p1b, 0x31);       /* x1 = fft_ip1: re, im */
        const __m256 dre = _mm256_fmadd_ps(b, d, _mm256_mul_ps(a, c));                                            /* fmaf(b, d, a * c) */
        const __m256 dim = _mm256_fmadd_ps(b, c, _mm256_xor_ps(_mm256_mul_ps(a, d), sign));                       /* fmaf(b, c, -(a * d)) */
        const __m256 ar = _mm256_and_ps(dre, absmask), ai = _mm256_and_ps(dim, absmask);
        const __m256 A = _mm256_blendv_ps(ar, ai, _mm256_cmp_ps(ar, ai, _CMP_LT_OQ));                             /* (ar < ai) ? ai : ar */
        const __m256 nr = _mm256_div_ps(dre, A), ni = _mm256_div_ps(dim, A);
        const __m256 vr = _mm256_mul_ps(_mm256_xor_ps(nr, sign), k127);                                           /* -(+nr) * 127 */
        const __m256 vi = _mm256_mul_ps(ni, k127);                                                                /* -(-ni) * 127 */
        const __m256i ir = _mm256_cvttps_epi32(vr), ii = _mm256_cvttps_epi32(vi);
        const __m128i r_lo = _mm_shuffle_epi8(_mm256_castsi256_si128(ir), pick), r_hi = _mm_shuffle_epi8(_mm256_extracti128_si256(ir, 1), pick);
        const __m128i i_lo = _mm_shuffle_epi8(_mm256_castsi256_si128(ii), pick), i_hi = _mm_shuffle_epi8(_mm256_extracti128_si256(ii, 1), pick);
        _mm_storel_epi64((__m128i *)(bits + i), _mm_unpacklo_epi32(r_lo, r_hi));
        _mm_storel_epi64((__m128i *)(bits + i + N), _mm_unpacklo_epi32(i_lo, i_hi));
    }
    return i;
}
#endif

/* ofdm_demodulator.cpp:842-889 */
DAB_HOT
void dab_dqpsk_demap(const dab_cf32 *fft_i, const dab_cf32 *fft_ip1, const int *mapper, int8_t *bits) {
    const int N = DAB_NB_DATA_CARRIERS, M = N / 2;
    int i0 = 0;
#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
    if (g_scalar_fft < 0) { const char *e = getenv("DAB_ORACLE_SCALAR_FFT"); g_scalar_fft = (e && e[0] == '1') || !(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")); }
    if (!g_scalar_fft) i0 = demap_avx2(fft_i, fft_ip1, mapper, bits, N);
#endif
    for (int i = i0; i < N; i++) {
        const int c = mapper[i];                                   /* :874 */
        const int k = (c < M) ? (c - M) : (c - M + 1);             /* carrier -768..-1,1..768 (:853-864) */
        const int bin = (DAB_NB_FFT + k) % DAB_NB_FFT;
        const dab_cf32 d = conj_mul(fft_i[bin], fft_ip1[bin]);     /* in1*conj(in0) :861, in1 = symbol i */
        const float ar = fabsf(d.re), ai = fabsf(d.im);
        const float A = (ar < ai) ? ai : ar;                       /* std::max :882 */
        const float nr = d.re / A, ni = d.im / A;                  /* :883 */
        bits[i]     = to_vbit(+nr);                                /* :886 */
        bits[i + N] = to_vbit(-ni);                                /* :887 */
    }
}

/* ofdm_demodulator.cpp:650-766 (single pipeline thread order) */
float dab_demod_frame(const dab_cf32 *frame, float f, const int *mapper,
                      int8_t *bits, dab_cf32 *cp_corr, float *cp_phase, dab_cf32 *fft_out) {
    ensure_tw();
    static __thread dab_cf32 sym[DAB_NB_SYMBOL_PERIOD];
    static __thread dab_cf32 X[2][DAB_NB_FFT];
    float total = 0.0f;
    for (int i = 0; i <= DAB_NB_FRAME_SYMBOLS; i++) {
        const dab_cf32 *src = frame + (size_t)i * DAB_NB_SYMBOL_PERIOD;     /* ofdm_frame_buffer.h:87-99 */
        const float dt0 = (float)(i * DAB_NB_SYMBOL_PERIOD) * f;            /* :675-676 */
        dab_apply_pll(src, sym, DAB_NB_SYMBOL_PERIOD, f, dt0);              /* :677 */
        if (i < DAB_NB_FRAME_SYMBOLS) {                                     /* :686-690 */
            const dab_cf32 c = dab_cp_correlation(sym);
            const float ph = dab_atan2f(c.im, c.re);
            if (cp_corr) cp_corr[i] = c;
            if (cp_phase) cp_phase[i] = ph;
            total += ph;
        }
        if (i == DAB_NB_FRAME_SYMBOLS && !fft_out) break;                   /* NULL FFT is display-only */
        dab_cf32 *cur = X[i & 1];
        fft2048_core(sym + DAB_NB_CYCLIC_PREFIX, cur, 0);                   /* :701-709 */
        if (fft_out) memcpy(fft_out + (size_t)i * DAB_NB_FFT, cur, sizeof(dab_cf32) * DAB_NB_FFT);
        if (i >= 1 && i < DAB_NB_FRAME_SYMBOLS)                             /* :728-739 */
            dab_dqpsk_demap(X[(i - 1) & 1], cur, mapper, bits + (size_t)(i - 1) * DAB_NB_SYM_BITS);
    }
    return total;
}

/* n back-to-back frames through dab_demod_frame (cpu_baseline timing loop: one GIL-free call per thread) */
void dab_demod_frames(const dab_cf32 *frames, size_t n_distinct, size_t n_total, float f, const int *mapper,
                      int8_t *bits_scratch, float *totals) {
    for (size_t k = 0; k < n_total; k++) {
        const float t = dab_demod_frame(frames + (k % n_distinct) * DAB_NB_FRAME_SAMPLES, f, mapper, bits_scratch, NULL, NULL, NULL);
        if (totals) totals[k] = t;
    }
}

/* ofdm_demodulator.cpp:829-840 */
float dab_fine_freq_add(float fine, float delta) {
    const float spacing = 1.0f / (float)DAB_NB_FFT;
    const float wrap = 0.5f * spacing * 1.01f;
    fine += delta;
    return fmodf(fine, wrap);
}

/* ofdm_demodulator.cpp:606-618 + :779-824 */
float dab_update_fine_freq(float fine, float total_phase_error) {
    const float TWO_PI = (float)M_PI * 2.0f;
    const float avg = total_phase_error / (float)DAB_NB_FRAME_SYMBOLS;
    const float spacing = 1.0f / (float)DAB_NB_FFT;
    const float err = spacing * avg / TWO_PI;
    const float delta = -0.9f * err;
    return dab_fine_freq_add(fine, delta);
}

/* ------------------------------------------------------------------------------------------ */
/* sync                                                                                         */
/* ------------------------------------------------------------------------------------------ */

void dab_sync_cfg_default(dab_sync_cfg *c) {         /* ofdm_demodulator.h:34-44 */
    c->fine_freq_update_beta = 0.9f;
    c->is_coarse_freq_correction = 1;
    c->max_coarse_freq_correction_norm = 0.5f;
    c->coarse_freq_slow_beta = 0.1f;
    c->impulse_peak_threshold_db = 20.0f;
    c->impulse_peak_distance_probability = 0.15f;
}

/* ofdm_demodulator.cpp:901-909 ; conj(in[i]) * in[i+1] == in[i+1] * conj(in[i]) */
static void relative_phase(const dab_cf32 *in, dab_cf32 *out) {
    for (int i = 0; i < DAB_NB_FFT - 1; i++) out[i] = conj_mul(in[i + 1], in[i]);
    out[DAB_NB_FFT - 1].re = 0.0f; out[DAB_NB_FFT - 1].im = 0.0f;
}

/* ofdm_demodulator.cpp:128-140 */
void dab_sync_refs(const dab_cf32 *prs_fft, dab_cf32 *prs_fft_conj, dab_cf32 *prs_time_ref) {
    dab_cf32 tmp[DAB_NB_FFT];
    for (int i = 0; i < DAB_NB_FFT; i++) { prs_fft_conj[i].re = prs_fft[i].re; prs_fft_conj[i].im = -prs_fft[i].im; }
    relative_phase(prs_fft, tmp);
    dab_ifft2048(tmp, prs_time_ref);
    for (int i = 0; i < DAB_NB_FFT; i++) prs_time_ref[i].im = -prs_time_ref[i].im;
}

/* 2048 values: leaf t sums elements t, t+256, ... sequentially, then the 256-leaf tree */
static float tree_sum_2048(const float *v) {
    float a[256];
    for (int t = 0; t < 256; t++) {
        float s = v[t];
        for (int j = 1; j < 8; j++) s += v[t + 256 * j];
        a[t] = s;
    }
    return tree256(a);
}

/* ofdm_demodulator.cpp:360-471 */
void dab_coarse_freq_sync(const dab_cf32 *prs_sym, const dab_cf32 *prs_time_ref, const dab_sync_cfg *cfg,
                          dab_sync_state *st, float *freq_response) {
    if (!cfg->is_coarse_freq_correction) { st->freq_coarse = 0.0f; return; }     /* :363-367 */
    const int N = DAB_NB_FFT, M = N / 2;
    dab_cf32 A[DAB_NB_FFT], B[DAB_NB_FFT];
    float resp[DAB_NB_FFT];
    dab_fft2048(prs_sym, A);                                    /* :377 */
    relative_phase(A, B);                                       /* :380 (out of place, same values) */
    dab_ifft2048(B, A);                                         /* :383 */
    for (int i = 0; i < N; i++) A[i] = cmul(A[i], prs_time_ref[i]);   /* :387-389 */
    dab_fft2048(A, B);                                          /* :392 */
    for (int i = 0; i < N; i++) resp[i] = dab_db20f(dab_cabsf(B[(i + M) % N]));  /* :911-920 */
    if (freq_response) memcpy(freq_response, resp, sizeof(resp));

    int max_off = (int)(cfg->max_coarse_freq_correction_norm * (float)N);        /* :399-402 */
    if (max_off < 0) max_off = 0;
    if (max_off > M) max_off = M;
    int max_index = -max_off;
    float max_value = resp[max_index + M];
    for (int i = -max_off; i <= max_off; i++) {                 /* :405-413 */
        const int idx = i + M;
        if (idx == N) continue;
        if (resp[idx] > max_value) { max_value = resp[idx]; max_index = i; }
    }
    int pidx[3]; float pmag[3];
    for (int j = 0; j < 3; j++) {                               /* :423-434 */
        int index = max_index - 1 + j;
        if (index < -max_off) index = -max_off;
        if (index > max_off) index = max_off;
        int fi = index + M;
        if (fi >= N) fi = N - 1;
        pidx[j] = fi - M;
        pmag[j] = dab_undb20f(resp[fi]);
    }
    float peak_sum = 0.0f, lerp = 0.0f;
    for (int j = 0; j < 3; j++) peak_sum += pmag[j];
    for (int j = 0; j < 3; j++) lerp += (float)pidx[j] * pmag[j] / peak_sum;     /* :438 */
    const float predicted = -lerp / (float)N;
    const float error = predicted - st->freq_coarse;
    const float large_thresh = 1.5f / (float)N;
    const int is_large = fabsf(error) > large_thresh;
    const int is_fast = is_large || !st->is_found_coarse;
    const float beta = is_fast ? 1.0f : cfg->coarse_freq_slow_beta;
    const float delta = beta * error;
    st->freq_coarse += delta;                                   /* :461 */
    st->is_found_coarse = 1;
    st->freq_fine = dab_fine_freq_add(st->freq_fine, -delta);   /* :467 */
}

/* ofdm_demodulator.cpp:473-548 (numeric part) */
int dab_fine_time_sync(const dab_cf32 *prs_sym, const dab_cf32 *prs_fft_conj, const dab_sync_cfg *cfg,
                       float freq_offset, int *offset, float *impulse_response) {
    const int N = DAB_NB_FFT;
    dab_cf32 A[DAB_NB_FFT], B[DAB_NB_FFT];
    float ir[DAB_NB_FFT];
    dab_apply_pll(prs_sym, A, N, freq_offset, 0.0f);            /* :481-482 */
    dab_fft2048(A, B);                                          /* :487 */
    for (int i = 0; i < N; i++) B[i] = cmul(B[i], prs_fft_conj[i]);   /* :488-490 */
    dab_ifft2048(B, A);                                         /* :493 */
    for (int i = 0; i < N; i++) ir[i] = dab_db20f(dab_cabsf(A[i]));   /* :494-498 */
    if (impulse_response) memcpy(impulse_response, ir, sizeof(ir));

    float max_value = ir[0];
    int max_index = 0;
    const float decay = 1.0f - cfg->impulse_peak_distance_probability;           /* :515 */
    for (int i = 0; i < N; i++) {                               /* :505-524 */
        const int dist = abs(DAB_NB_CYCLIC_PREFIX - i);
        const float norm_dist = (float)dist / (float)DAB_NB_SYMBOL_PERIOD;
        const float prob = 1.0f - decay * norm_dist;
        const float w = prob * ir[i];
        if (w > max_value) { max_value = w; max_index = i; }
    }
    const float avg = tree_sum_2048(ir) / (float)N;            /* :519,:525 ; summation order = contract tree */
    if ((max_value - avg) < cfg->impulse_peak_threshold_db) return 0;            /* :529 */
    *offset = max_index - DAB_NB_CYCLIC_PREFIX;                 /* :536 */
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Transmission modes II, III, IV: the same demodulator over the other geometries  */
/* (src/ofdm/dab_ofdm_params_ref.cpp:11-60).  Mode I goes through the functions above; the         */
/* generic ones reproduce them exactly for mode I (tests/test_oracle_modes.py).                     */
/* ------------------------------------------------------------------------------------------ */

int dab_ofdm_geometry_get(int mode, dab_ofdm_geometry *g) {
    static const int T[5][5] = { {0, 0, 0, 0, 0}, {76, 2552, 2656, 2048, 1536}, {76, 638, 664, 512, 384},
                                 {153, 319, 345, 256, 192}, {76, 1276, 1328, 1024, 768} };
    if (mode < 1 || mode > 4) return -1;
    g->mode = mode;
    g->nb_frame_symbols = T[mode][0]; g->nb_symbol_period = T[mode][1]; g->nb_null_period = T[mode][2];
    g->nb_fft = T[mode][3]; g->nb_carriers = T[mode][4];
    g->nb_cp = g->nb_symbol_period - g->nb_fft;
    g->nb_frame_samples = g->nb_frame_symbols * g->nb_symbol_period + g->nb_null_period;
    g->nb_sym_bits = 2 * g->nb_carriers;
    g->nb_frame_bits = (g->nb_frame_symbols - 1) * g->nb_sym_bits;
    return 0;
}

/* get_DAB_mapper_ref for any (nb_fft, nb_carriers), src/ofdm/dab_mapper_ref.cpp:10-51 */
void dab_mapper_n(int nb_fft, int nb_carriers, int *out) {
    const int N = nb_fft, dc = N / 2, lo = dc - nb_carriers / 2, hi = dc + nb_carriers / 2;
    int v = 0, n = 0;
    for (int i = 0; i < N; i++) {
        if (i > 0) v = (13 * v + N / 4 - 1) % N;
        if (v < lo || v > hi || v == dc) continue;
        out[n++] = (v < dc) ? (v - lo) : (v - lo - 1);
    }
}

/* FFT contract for n = r1 * 8^k (2048 = 4.8.8.8, 1024 = 2.8.8.8, 512 = 8.8.8, 256 = 4.8.8): Stockham autosort, decimation
 * in frequency, one radix-r1 pass then radix-8 passes; twiddle w_n^m = g_tw[m * 2048 / n]; output 0 of a butterfly is not
 * multiplied, the last pass has no twiddles */
void dab_fft_n(int n, const dab_cf32 *in, dab_cf32 *out, int inverse) {
    ensure_tw();
    dab_cf32 bufs[2][DAB_NB_FFT];
    int radix[4], n_pass = 0;
    { int rem = n; if (n == 2048 || n == 256) { radix[n_pass++] = 4; rem /= 4; } else if (n == 1024) { radix[n_pass++] = 2; rem /= 2; }
      while (rem > 1) { radix[n_pass++] = 8; rem /= 8; } }
    const dab_cf32 *src = in;
    int cur_n = n, s = 1;
    for (int ps = 0; ps < n_pass; ps++) {
        const int r = radix[ps], m = cur_n / r, last = (ps == n_pass - 1), first = (ps == 0);
        dab_cf32 *dst = last ? out : bufs[ps & 1];
        const int tw_step = DAB_NB_FFT / cur_n;
        for (int p = 0; p < m; p++)
            for (int q = 0; q < s; q++) {
                dab_cf32 a[8], b[8];
                for (int j = 0; j < r; j++) {
                    a[j] = src[q + s * (p + m * j)];
                    if (first && inverse) a[j].im = -a[j].im;
                }
                if (r == 8) dft8(a, b);
                else if (r == 4) dft4(a, b);
                else { b[0] = cadd(a[0], a[1]); b[1] = csub(a[0], a[1]); }
                for (int k = 0; k < r; k++) {
                    dab_cf32 v = (k == 0 || last) ? b[k] : cmul(b[k], g_tw[tw_step * p * k]);
                    if (last && inverse) v.im = -v.im;
                    dst[q + s * (r * p + k)] = v;
                }
            }
        src = dst;
        cur_n = m;
        s *= r;
    }
}

/* cyclic-prefix correlation: mode I keeps its 256-leaf tree (above); the other modes take one leaf per sample,
 * L = the power of two >= nb_cp, and halve with strides L/2 .. 1 */
dab_cf32 dab_cp_correlation_n(const dab_cf32 *sym, int nb_fft, int nb_cp) {
    if (nb_fft == DAB_NB_FFT) return dab_cp_correlation(sym);
    float pr[256], pi[256];
    int L = 1;
    while (L < nb_cp) L <<= 1;
    for (int j = 0; j < L; j++) {
        if (j < nb_cp) { const dab_cf32 p = conj_mul(sym[nb_fft + j], sym[j]); pr[j] = p.re; pi[j] = p.im; }
        else { pr[j] = 0.0f; pi[j] = 0.0f; }
    }
    for (int h = L / 2; h >= 1; h >>= 1)
        for (int i = 0; i < h; i++) { pr[i] += pr[i + h]; pi[i] += pi[i + h]; }
    dab_cf32 r = { pr[0], pi[0] };
    return r;
}

float dab_demod_frame_mode(int mode, const dab_cf32 *frame, float f, const int *mapper, int8_t *bits, dab_cf32 *cp_corr,
                           float *cp_phase, dab_cf32 *fft_out) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return 0.0f;
    dab_cf32 sym[DAB_NB_SYMBOL_PERIOD];
    static __thread dab_cf32 X[2][DAB_NB_FFT];
    const int N = g.nb_fft, NC = g.nb_carriers, M = NC / 2;
    float total = 0.0f;
    for (int i = 0; i <= g.nb_frame_symbols; i++) {
        const dab_cf32 *src = frame + (size_t)i * g.nb_symbol_period;
        const float dt0 = (float)(i * g.nb_symbol_period) * f;
        dab_apply_pll(src, sym, (size_t)g.nb_symbol_period, f, dt0);
        if (i < g.nb_frame_symbols) {
            const dab_cf32 c = dab_cp_correlation_n(sym, N, g.nb_cp);
            const float ph = dab_atan2f(c.im, c.re);
            if (cp_corr) cp_corr[i] = c;
            if (cp_phase) cp_phase[i] = ph;
            total += ph;
        }
        if (i == g.nb_frame_symbols && !fft_out) break;
        dab_cf32 *cur = X[i & 1];
        dab_fft_n(N, sym + g.nb_cp, cur, 0);
        if (fft_out) memcpy(fft_out + (size_t)i * N, cur, sizeof(dab_cf32) * (size_t)N);
        if (i >= 1 && i < g.nb_frame_symbols) {
            const dab_cf32 *prev = X[(i - 1) & 1];
            int8_t *o = bits + (size_t)(i - 1) * g.nb_sym_bits;
            for (int n = 0; n < NC; n++) {
                const int c = mapper[n];
                const int k = (c < M) ? (c - M) : (c - M + 1);
                const int bin = (N + k) % N;
                const dab_cf32 d = conj_mul(prev[bin], cur[bin]);
                const float ar = fabsf(d.re), ai = fabsf(d.im);
                const float A = (ar < ai) ? ai : ar;
                o[n] = to_vbit(+(d.re / A));
                o[n + NC] = to_vbit(-(d.im / A));
            }
        }
    }
    return total;
}

/* ofdm_demodulator.cpp:606-618 + :779-840 for any mode */
float dab_update_fine_freq_mode(int mode, float fine, float total_phase_error, float beta) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return fine;
    const float TWO_PI = (float)M_PI * 2.0f;
    const float avg = total_phase_error / (float)g.nb_frame_symbols;
    const float spacing = 1.0f / (float)g.nb_fft;
    const float err = spacing * avg / TWO_PI;
    const float delta = -beta * err;
    const float wrap = 0.5f * spacing * 1.01f;
    fine += delta;
    return fmodf(fine, wrap);
}

/* ---- phase reference symbol and PRS synchronisation for any mode ---- */

/* ETSI EN 300 401 clause 14.3.2 (tables for modes II-IV: docs/DAB_implementation_in_SDR_detailed.pdf appendix B; the same
 * data as src/ofdm/dab_prs_ref.cpp:25-120): per 32-carrier block the row i of the h table and the offset n */
static const signed char PRS_ROWS_I[4][48] = {
    { 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3,  0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1 },
    { 0,1,2,3,0,1,  2,1,0,3,2,1 },
    { 0,1,2,  3,2,1 },
    { 0,1,2,3, 0,1,2,3, 0,1,2,3,  0,3,2,1, 0,3,2,1, 0,3,2,1 },
};
static const signed char PRS_ROWS_N[4][48] = {
    { 1,2,0,1, 3,2,2,3, 2,1,2,3, 1,2,3,3, 2,2,2,1, 1,3,1,2,  3,1,1,1, 2,2,1,0, 2,2,3,3, 0,2,1,3, 3,3,3,0, 3,0,1,1 },
    { 2,3,2,2,1,2,  0,2,2,1,0,3 },
    { 2,3,0,  2,2,2 },
    { 0,1,1,2, 2,2,0,3, 3,1,3,2,  0,1,0,2, 0,1,2,2, 2,1,3,0 },
};

int dab_get_prs_fft_mode(int mode, dab_cf32 *prs) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return -1;
    const int N = g.nb_fft, rows = g.nb_carriers / 32, half = rows / 2;
    memset(prs, 0, sizeof(dab_cf32) * (size_t)N);
    for (int row = 0; row < rows; row++) {
        const int k_min = (row < half) ? (-g.nb_carriers / 2 + 32 * row) : (1 + 32 * (row - half));
        for (int j = 0; j < 32; j++) {
            const int k = k_min + j;
            const int h = PRS_H[(int)PRS_ROWS_I[mode - 1][row]][j];
            const float phi = (float)M_PI / 2.0f * (float)(h + PRS_ROWS_N[mode - 1][row]);
            const int bin = (k < 0) ? (N + k) : k;
            prs[bin].re = cosf(phi);
            prs[bin].im = sinf(phi);
        }
    }
    return 0;
}

static void relative_phase_n(int n, const dab_cf32 *in, dab_cf32 *out) {
    for (int i = 0; i < n - 1; i++) out[i] = conj_mul(in[i + 1], in[i]);
    out[n - 1].re = 0.0f; out[n - 1].im = 0.0f;
}

void dab_sync_refs_mode(int mode, const dab_cf32 *prs_fft, dab_cf32 *prs_fft_conj, dab_cf32 *prs_time_ref) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return;
    const int N = g.nb_fft;
    dab_cf32 tmp[DAB_NB_FFT];
    for (int i = 0; i < N; i++) { prs_fft_conj[i].re = prs_fft[i].re; prs_fft_conj[i].im = -prs_fft[i].im; }
    relative_phase_n(N, prs_fft, tmp);
    dab_fft_n(N, tmp, prs_time_ref, 1);
    for (int i = 0; i < N; i++) prs_time_ref[i].im = -prs_time_ref[i].im;
}

/* leaf t sums elements t, t+256, ... sequentially, then the 256-leaf tree (n >= 256) */
static float tree_sum_n(int n, const float *v) {
    float a[256];
    for (int t = 0; t < 256; t++) {
        float s = v[t];
        for (int j = 1; j < n / 256; j++) s += v[t + 256 * j];
        a[t] = s;
    }
    return tree256(a);
}

static float fine_freq_add_n(int n, float fine, float delta) {
    const float spacing = 1.0f / (float)n;
    const float wrap = 0.5f * spacing * 1.01f;
    fine += delta;
    return fmodf(fine, wrap);
}

void dab_coarse_freq_sync_mode(int mode, const dab_cf32 *prs_sym, const dab_cf32 *prs_time_ref, const dab_sync_cfg *cfg,
                               dab_sync_state *st, float *freq_response) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return;
    if (!cfg->is_coarse_freq_correction) { st->freq_coarse = 0.0f; return; }
    const int N = g.nb_fft, M = N / 2;
    dab_cf32 A[DAB_NB_FFT], B[DAB_NB_FFT];
    float resp[DAB_NB_FFT];
    dab_fft_n(N, prs_sym, A, 0);
    relative_phase_n(N, A, B);
    dab_fft_n(N, B, A, 1);
    for (int i = 0; i < N; i++) A[i] = cmul(A[i], prs_time_ref[i]);
    dab_fft_n(N, A, B, 0);
    for (int i = 0; i < N; i++) resp[i] = dab_db20f(dab_cabsf(B[(i + M) % N]));
    if (freq_response) memcpy(freq_response, resp, sizeof(float) * (size_t)N);
    int max_off = (int)(cfg->max_coarse_freq_correction_norm * (float)N);
    if (max_off < 0) max_off = 0;
    if (max_off > M) max_off = M;
    int max_index = -max_off;
    float max_value = resp[max_index + M];
    for (int i = -max_off; i <= max_off; i++) {
        const int idx = i + M;
        if (idx == N) continue;
        if (resp[idx] > max_value) { max_value = resp[idx]; max_index = i; }
    }
    int pidx[3]; float pmag[3];
    for (int j = 0; j < 3; j++) {
        int index = max_index - 1 + j;
        if (index < -max_off) index = -max_off;
        if (index > max_off) index = max_off;
        int fi = index + M;
        if (fi >= N) fi = N - 1;
        pidx[j] = fi - M;
        pmag[j] = dab_undb20f(resp[fi]);
    }
    float peak_sum = 0.0f, lerp = 0.0f;
    for (int j = 0; j < 3; j++) peak_sum += pmag[j];
    for (int j = 0; j < 3; j++) lerp += (float)pidx[j] * pmag[j] / peak_sum;
    const float predicted = -lerp / (float)N;
    const float error = predicted - st->freq_coarse;
    const float large_thresh = 1.5f / (float)N;
    const int is_large = fabsf(error) > large_thresh;
    const int is_fast = is_large || !st->is_found_coarse;
    const float beta = is_fast ? 1.0f : cfg->coarse_freq_slow_beta;
    const float delta = beta * error;
    st->freq_coarse += delta;
    st->is_found_coarse = 1;
    st->freq_fine = fine_freq_add_n(N, st->freq_fine, -delta);
}

int dab_fine_time_sync_mode(int mode, const dab_cf32 *prs_sym, const dab_cf32 *prs_fft_conj, const dab_sync_cfg *cfg,
                            float freq_offset, int *offset, float *impulse_response) {
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return 0;
    const int N = g.nb_fft;
    dab_cf32 A[DAB_NB_FFT], B[DAB_NB_FFT];
    float ir[DAB_NB_FFT];
    dab_apply_pll(prs_sym, A, (size_t)N, freq_offset, 0.0f);
    dab_fft_n(N, A, B, 0);
    for (int i = 0; i < N; i++) B[i] = cmul(B[i], prs_fft_conj[i]);
    dab_fft_n(N, B, A, 1);
    for (int i = 0; i < N; i++) ir[i] = dab_db20f(dab_cabsf(A[i]));
    if (impulse_response) memcpy(impulse_response, ir, sizeof(float) * (size_t)N);
    float max_value = ir[0];
    int max_index = 0;
    const float decay = 1.0f - cfg->impulse_peak_distance_probability;
    for (int i = 0; i < N; i++) {
        const int dist = abs(g.nb_cp - i);
        const float norm_dist = (float)dist / (float)g.nb_symbol_period;
        const float prob = 1.0f - decay * norm_dist;
        const float w = prob * ir[i];
        if (w > max_value) { max_value = w; max_index = i; }
    }
    const float avg = tree_sum_n(N, ir) / (float)N;
    if ((max_value - avg) < cfg->impulse_peak_threshold_db) return 0;
    *offset = max_index - g.nb_cp;
    return 1;
}
