// ofdm_sync.hip -- PRS-based synchronisation on the device, one 256-thread workgroup per stream:
//   coarse frequency sync  (ref: OFDM_Demod::RunCoarseFreqSync, src/ofdm/ofdm_demodulator.cpp:360-471)
//   fine time sync          (ref: OFDM_Demod::RunFineTimeSync,   src/ofdm/ofdm_demodulator.cpp:473-548)
// Five 2048-point transforms per frame, all through the same butterflies as the symbol kernel (ofdm_device.h), the
// dB / argmax / lerp tails with the deterministic log/exp of the arithmetic contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabgpu_internal.h"
#include "ofdm_device.h"

namespace dabgpu {

// ---- deterministic scalar functions (operation-for-operation the oracle's dab_cabsf / dab_db20f / dab_undb20f) ----
__device__ __forceinline__ float cabs_det(f2 v) { return __builtin_sqrtf(fma_(v.x, v.x, v.y * v.y)); }

__device__ __forceinline__ float db20_det(float m) {
    if (!(m > 0.0f)) return -__builtin_inff();
    float e_adj = 0.0f;
    if (m < 1.17549435e-38f) { m = m * 16777216.0f; e_adj = -24.0f; }
    uint32_t u = __float_as_uint(m);
    int e = (int)((u >> 23) & 0xFFu) - 126;
    u = (u & 0x007FFFFFu) | 0x3F000000u;
    float f = __uint_as_float(u);
    if (f < 0.707106769084930420f) { e -= 1; f = f + f; }
    const float x = f - 1.0f;
    const float z = x * x;
    float p = fma_(7.0376836292e-2f, x, -1.1514610310e-1f);
    p = fma_(p, x, 1.1676998740e-1f);
    p = fma_(p, x, -1.2420140846e-1f);
    p = fma_(p, x, 1.4249322787e-1f);
    p = fma_(p, x, -1.6668057665e-1f);
    p = fma_(p, x, 2.0000714765e-1f);
    p = fma_(p, x, -2.4999993993e-1f);
    p = fma_(p, x, 3.3333331174e-1f);
    float y = (p * x) * z;
    y = fma_(-0.5f, z, y);
    const float fe = (float)e + e_adj;
    const float ln = fma_(fe, 0.693147182464599609f, x + y);
    return ln * 8.68588924407958984f;
}

__device__ __forceinline__ float undb20_det(float db) {
    float x = db * 0.166096404194831848f;
    if (x > 127.0f) return __builtin_inff();
    if (!(x > -149.0f)) return 0.0f;
    float n = __builtin_floorf(x);
    x = x - n;
    if (x > 0.5f) { n += 1.0f; x -= 1.0f; }
    float p = fma_(1.535336188319500e-4f, x, 1.339887440266574e-3f);
    p = fma_(p, x, 9.618437357674640e-3f);
    p = fma_(p, x, 5.550332471162809e-2f);
    p = fma_(p, x, 2.402264791363012e-1f);
    p = fma_(p, x, 6.931472028550421e-1f);
    p = fma_(p, x, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// UpdateFineFrequencyOffset (ofdm_demodulator.cpp:829-840)
__device__ __forceinline__ float fine_freq_add(float fine, float delta, int n_fft) {
    const float spacing = 1.0f / (float)n_fft;
    const float wrap = 0.5f * spacing * 1.01f;
    fine += delta;
    return fmodf(fine, wrap);
}

// ---- 2048-point transform between natural-order LDS arrays (x -> y), same pass structure as ofdm_demod_kernel ----
// conj_io: inverse transform as conj(FFT(conj(x))), unnormalised like FFTW_BACKWARD
// x, bufA and y may all be ONE array (in place): a thread writes bufA at exactly the eight positions it has read x at, and the results
// are written behind a barrier
// the 20 twiddles a thread needs in a 2048-point transform depend on its index only: a kernel that runs several transforms loads them once
struct Fft2048Tw { f2 p1[6], p2[7], p3[7]; };
__device__ __forceinline__ Fft2048Tw fft2048_twiddles(const f2* __restrict__ tw) {
    const int t = threadIdx.x, lane = t & 63, la = lane & 7;
    Fft2048Tw w;
#pragma unroll
    for (int k = 1; k <= 3; k++) { w.p1[k - 1] = tw[(2 * t) * k]; w.p1[2 + k] = tw[(2 * t + 1) * k]; }
#pragma unroll
    for (int k = 1; k < 8; k++) { w.p2[k - 1] = tw[4 * lane * k]; w.p3[k - 1] = tw[32 * la * k]; }
    return w;
}

// x, y and the exchange array are ONE array of 4 x WAVE_PATCH elements (in place): input and output in natural order in its first 2048
// elements; between them the radix-4 outputs sit as four 512-element blocks WAVE_PATCH apart, one per wave's 512-point problem, and a
// wave's transpose patch aliases its own block (the wave has read its 8 inputs per lane before it writes the patch; one wave's LDS
// instructions execute in order) -- the layout of the symbol kernel (ofdm_demod.hip).  Barriers: inputs read / blocks written / results written.
__device__ __forceinline__ void fft2048_lds(f2* A, const Fft2048Tw& w, bool conj_io) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int la = lane & 7, lb = lane >> 3;
    f2* patch = A + wave * WAVE_PATCH;
    f2 a[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { a[j] = A[2 * t + 512 * j]; a[4 + j] = A[2 * t + 1 + 512 * j]; }
    if (conj_io) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j].y = -a[j].y;
    }
    __syncthreads();                                       // (the blocks below do not sit where the inputs did)
    {
        f2 b0, b1, b2, b3, c0, c1, c2, c3;
        dft4(a[0], a[1], a[2], a[3], b0, b1, b2, b3);
        dft4(a[4], a[5], a[6], a[7], c0, c1, c2, c3);
        b1 = cmul(b1, w.p1[0]); b2 = cmul(b2, w.p1[1]); b3 = cmul(b3, w.p1[2]);
        c1 = cmul(c1, w.p1[3]); c2 = cmul(c2, w.p1[4]); c3 = cmul(c3, w.p1[5]);
        A[2 * t] = b0;                       A[2 * t + 1] = c0;
        A[2 * t + WAVE_PATCH] = b1;          A[2 * t + 1 + WAVE_PATCH] = c1;
        A[2 * t + 2 * WAVE_PATCH] = b2;      A[2 * t + 1 + 2 * WAVE_PATCH] = c2;
        A[2 * t + 3 * WAVE_PATCH] = b3;      A[2 * t + 1 + 3 * WAVE_PATCH] = c3;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[lane + 64 * j];
    dft8(a);
    patch[lane] = a[0];
#pragma unroll
    for (int k = 1; k < 8; k++) patch[lane + 72 * k] = cmul(a[k], w.p2[k - 1]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[la + 72 * lb + 8 * j];
    wave_lds_fence();
    dft8(a);
    patch[la + 72 * lb] = a[0];
#pragma unroll
    for (int k = 1; k < 8; k++) patch[la + 72 * lb + 9 * k] = cmul(a[k], w.p3[k - 1]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[9 * la + 72 * lb + j];
    wave_lds_fence();
    dft8(a);
    const int Kb = wave + 4 * lb + 32 * la;
    __syncthreads();                                       // every wave has taken its block out of the array before anybody writes a result
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f2 v = a[k];
        if (conj_io) v.y = -v.y;
        A[Kb + 256 * k] = v;
    }
    __syncthreads();
}

// transform of any supported length between natural-order LDS arrays: the register-resident 2048-point version above, or
// Stockham passes r1 x 8 x 8 [x 8] alternating between `tmp` and `y` so that the last pass lands in `y`
__device__ __forceinline__ void fft_lds(int N, f2* x, f2* y, f2* tmp, const f2* __restrict__ tw, const Fft2048Tw& w2048, bool conj_io) {
    if (N == NB_FFT) { fft2048_lds(x, w2048, conj_io); return; }      // (x == y == tmp: in place)
    const int t = threadIdx.x;
    if (conj_io) { for (int i = t; i < N; i += 256) x[i].y = -x[i].y; __syncthreads(); }
    const int r1 = (N == 256) ? 4 : (N == 1024 ? 2 : 8);
    const int n_pass = (N == 1024) ? 4 : 3;
    const f2* src = x;
    int cur_n = N, s = 1;
    for (int ps = 0; ps < n_pass; ps++) {
        const int r = (ps == 0) ? r1 : 8;
        const bool last = (ps == n_pass - 1);
        f2* dst = (((n_pass - 1 - ps) & 1) == 0) ? y : tmp;
        if (r == 8) stockham_pass<8>(src, dst, N, cur_n, s, last, tw, t, 256);
        else if (r == 4) stockham_pass<4>(src, dst, N, cur_n, s, last, tw, t, 256);
        else stockham_pass<2>(src, dst, N, cur_n, s, last, tw, t, 256);
        __syncthreads();
        src = dst; cur_n /= r; s *= r;
    }
    if (conj_io) { for (int i = t; i < N; i += 256) y[i].y = -y[i].y; __syncthreads(); }
}

// LDS of a synchroniser workgroup.
//   N = 2048 (mode I): ONE array A of 4 x WAVE_PATCH elements -- every transform runs in place in it (fft2048_lds), the element-wise steps
//                      between transforms pass through registers, the dB response is written over its first 8 KB behind a barrier:
//                      18.5 KB (dabgpu_sync_lds_bytes; P is not allocated), five to eight workgroups per CU.  (Round 3: 34.6 KB with the
//                      patches and the response in a second region, four per CU; before: 75.8 KB, two per CU.)
//   N <= 1024:         x = A, y = A + 1024 (Stockham passes ping-pong), tmp = P, R = P + 1024
struct SyncLds {
    f2 A[4 * WAVE_PATCH];
    float redv[4];
    int redi[4];
    float reds[4];
    f2 P[NB_FFT];                    // modes II-IV only (last member: a mode I launch does not allocate it)
    __device__ f2* x() { return A; }
    __device__ f2* y(int N) { return N == NB_FFT ? A : A + 1024; }
    __device__ f2* tmp(int N) { return N == NB_FFT ? A : P; }
    __device__ float* R(int N) { return reinterpret_cast<float*>(N == NB_FFT ? A : P + 1024); }
};
__host__ __device__ inline size_t dabgpu_sync_lds_bytes(int n_fft) { return n_fft == NB_FFT ? offsetof(SyncLds, P) : sizeof(SyncLds); }

// (value, index) reduction with "largest value, ties -> lowest index" == first maximum of a sequential strict-> scan
__device__ __forceinline__ void argmax_reduce(float& v, int& i, SyncLds* S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float v2 = __shfl_xor(v, off);
        const int i2 = __shfl_xor(i, off);
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
    }
    if (lane == 0) { S->redv[wave] = v; S->redi[wave] = i; }
    __syncthreads();
    v = S->redv[0]; i = S->redi[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        const float v2 = S->redv[w]; const int i2 = S->redi[w];
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
    }
    __syncthreads();
}

// constructor-time reference of the coarse sync (ofdm_demodulator.cpp:134-140): conj(IFFT(relative_phase(PRS)))
__global__ __launch_bounds__(256)
void sync_init_kernel(const f2* __restrict__ prs, const f2* __restrict__ tw, f2* __restrict__ prs_time_ref, int N) {
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    SyncLds* S = reinterpret_cast<SyncLds*>(ssm);
    const int t = threadIdx.x;
    Fft2048Tw w2048 = {};
    if (N == NB_FFT) w2048 = fft2048_twiddles(tw);
    for (int i = t; i < N; i += 256)
        S->x()[i] = (i < N - 1) ? conj_mul(prs[i + 1], prs[i]) : mk2(0.0f, 0.0f);         // CalculateRelativePhase :901-909
    __syncthreads();
    fft_lds(N, S->x(), S->y(N), S->tmp(N), tw, w2048, true);
    for (int i = t; i < N; i += 256) prs_time_ref[i] = mk2(S->y(N)[i].x, -S->y(N)[i].y);
}

__global__ __launch_bounds__(256)
void ofdm_sync_kernel(const f2* __restrict__ prs_syms, size_t stride_samples, int n_streams, dabgpu_sync_cfg cfg,
                      dabgpu_sync_state* __restrict__ states, float* __restrict__ impulse_out, float* __restrict__ freq_out,
                      const f2* __restrict__ tw, const f2* __restrict__ prs_fft, const f2* __restrict__ prs_time_ref,
                      const int* __restrict__ active, ModeGeom g)
{
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    SyncLds* S = reinterpret_cast<SyncLds*>(ssm);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int sidx = blockIdx.x;
    if (sidx >= n_streams) return;
    if (active != nullptr && active[sidx] == 0) return;        // stream banks: only streams whose PRS window just filled
    const f2* prs_sym = prs_syms + (size_t)sidx * stride_samples;
    dabgpu_sync_state st = states[sidx];
    const int N = g.n_fft, M = N / 2;
    Fft2048Tw w2048 = {};                                    // (mode I: the five transforms share their twiddles)
    if (N == NB_FFT) w2048 = fft2048_twiddles(tw);

    // ================= coarse frequency sync (:360-471) =================
    f2* const X = S->x();
    f2* const Y = S->y(N);
    f2* const T = S->tmp(N);
    float* const R = S->R(N);
    if (cfg.is_coarse_freq_correction) {
        for (int i = t; i < N; i += 256) X[i] = prs_sym[i];
        __syncthreads();
        fft_lds(N, X, Y, T, tw, w2048, false);                                               // :377
        {   // X[i] = Y[i + 1] * conj(Y[i]) (:380) -- X and Y may be one array: through registers, behind a barrier
            f2 q[NB_FFT / 256];
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) {
                const int i = t + 256 * j;
                q[j] = (i < N - 1) ? conj_mul(Y[i + 1], Y[i]) : mk2(0.0f, 0.0f);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; if (i < N) X[i] = q[j]; }
        }
        __syncthreads();
        fft_lds(N, X, Y, T, tw, w2048, true);                                                // :383
        {   // (same index in and out; X and Y may be one array or two)
            f2 q[NB_FFT / 256];
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; if (i < N) q[j] = cmul(Y[i], prs_time_ref[i]); }   // :387-389
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; if (i < N) X[i] = q[j]; }
        }
        __syncthreads();
        fft_lds(N, X, Y, T, tw, w2048, false);                                               // :392
        {   // :911-920 -- R may lie over Y (mode I): through registers, behind a barrier
            float r[NB_FFT / 256];
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; r[j] = (i < N) ? db20_det(cabs_det(Y[(i + M) % N])) : 0.0f; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NB_FFT / 256; j++) {
                const int i = t + 256 * j;
                if (i < N) { R[i] = r[j]; if (freq_out) freq_out[(size_t)sidx * N + i] = r[j]; }
            }
        }
        __syncthreads();
        int max_off = (int)(cfg.max_coarse_freq_correction_norm * (float)N);                // :399-402
        if (max_off < 0) max_off = 0;
        if (max_off > M) max_off = M;
        float bv = -__builtin_inff(); int bi = 0x7FFFFFFF;
        for (int idx = t; idx < N; idx += 256) {                                            // :405-413, idx == N never occurs
            const int i = idx - M;
            if (i < -max_off || i > max_off) continue;
            const float v = R[idx];
            if (v > bv || (v == bv && idx < bi)) { bv = v; bi = idx; }
        }
        argmax_reduce(bv, bi, S);
        if (t == 0) {
            const int max_index = (bi == 0x7FFFFFFF) ? -max_off : (bi - M);
            int pidx[3]; float pmag[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {                                                   // :423-434
                int index = max_index - 1 + j;
                if (index < -max_off) index = -max_off;
                if (index > max_off) index = max_off;
                int fi = index + M;
                if (fi >= N) fi = N - 1;
                pidx[j] = fi - M;
                pmag[j] = undb20_det(R[fi]);
            }
            float peak_sum = 0.0f, lerp = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; j++) peak_sum += pmag[j];
#pragma unroll
            for (int j = 0; j < 3; j++) lerp += (float)pidx[j] * pmag[j] / peak_sum;        // :438
            const float predicted = -lerp / (float)N;
            const float error = predicted - st.freq_coarse;
            const float large_thresh = 1.5f / (float)N;
            const bool is_large = __builtin_fabsf(error) > large_thresh;
            const bool is_fast = is_large || !st.is_found_coarse;
            const float beta = is_fast ? 1.0f : cfg.coarse_freq_slow_beta;
            const float delta = beta * error;
            st.freq_coarse += delta;                                                        // :461
            st.is_found_coarse = 1;
            st.freq_fine = fine_freq_add(st.freq_fine, -delta, N);                          // :467
            S->reds[0] = st.freq_coarse; S->reds[1] = st.freq_fine;
        }
        __syncthreads();
        st.freq_coarse = S->reds[0]; st.freq_fine = S->reds[1]; st.is_found_coarse = 1;
        __syncthreads();
    } else {
        st.freq_coarse = 0.0f;                                                              // :363-367
    }

    // ================= fine time sync (:473-548) =================
    const float f = st.freq_coarse + st.freq_fine;                                          // :480
    for (int i = t; i < N; i += 256) {                                                      // :481-482, apply_pll.cpp:81-117
        const int k = i & 3;
        const float ss = (float)k * f;
        const float base = 0.0f + (float)(i & ~3) * f;
        X[i] = pll1(prs_sym[i], base, mk2(ss + 0.25f, ss));
    }
    __syncthreads();
    fft_lds(N, X, Y, T, tw, w2048, false);                                                   // :487
    {
        f2 q[NB_FFT / 256];
#pragma unroll
        for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; if (i < N) q[j] = cmul(Y[i], mk2(prs_fft[i].x, -prs_fft[i].y)); }   // :488-490
#pragma unroll
        for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; if (i < N) X[i] = q[j]; }
    }
    __syncthreads();
    fft_lds(N, X, Y, T, tw, w2048, true);                                                    // :493
    {   // :494-498 (through registers, as above)
        float r[NB_FFT / 256];
#pragma unroll
        for (int j = 0; j < NB_FFT / 256; j++) { const int i = t + 256 * j; r[j] = (i < N) ? db20_det(cabs_det(Y[i])) : 0.0f; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB_FFT / 256; j++) {
            const int i = t + 256 * j;
            if (i < N) { R[i] = r[j]; if (impulse_out) impulse_out[(size_t)sidx * N + i] = r[j]; }
        }
    }
    __syncthreads();
    // weighted arg-max (:505-524) and mean of the dB response with the contract's 256-leaf tree
    const float decay = 1.0f - cfg.impulse_peak_distance_probability;
    float bv = -__builtin_inff(); int bi = 0x7FFFFFFF;
    float leaf = 0.0f;
    for (int j = 0; j < N / 256; j++) {
        const int i = t + 256 * j;
        const float r = R[i];
        leaf = (j == 0) ? r : (leaf + r);
        const int dist = abs(g.n_cp - i);
        const float norm_dist = (float)dist / (float)g.period;
        const float prob = 1.0f - decay * norm_dist;
        const float w = prob * r;
        if (w > bv || (w == bv && i < bi)) { bv = w; bi = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) leaf += __shfl_xor(leaf, off);
    if (lane == 0) S->reds[wave] = leaf;
    argmax_reduce(bv, bi, S);                                                               // (contains the barriers)
    if (t == 0) {
        const float total = (S->reds[0] + S->reds[1]) + (S->reds[2] + S->reds[3]);
        const float avg = total / (float)N;
        const float r0 = R[0];                                                           // scan starts from the unweighted [0] (:503)
        float max_value = r0; int max_index = 0;
        if (bv > r0) { max_value = bv; max_index = bi; }
        const bool valid = !((max_value - avg) < cfg.impulse_peak_threshold_db);            // :529
        st.sync_valid = valid ? 1 : 0;
        if (valid) st.fine_time_offset = max_index - g.n_cp;                               // :536
        states[sidx] = st;
    }
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_sync_init(const float* d_prs, const float* d_tw, float* d_prs_time_ref, int n_fft, hipStream_t stream) {
    using namespace dabgpu;
    hipLaunchKernelGGL(sync_init_kernel, dim3(1), dim3(256), dabgpu_sync_lds_bytes(n_fft), stream,
                       reinterpret_cast<const f2*>(d_prs), reinterpret_cast<const f2*>(d_tw), reinterpret_cast<f2*>(d_prs_time_ref), n_fft);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_sync(const float* d_prs_syms, size_t stride_samples, int n_streams, const dabgpu_sync_cfg* cfg,
                                         dabgpu_sync_state* d_states, float* d_impulse, float* d_freq, const float* d_tw,
                                         const float* d_prs, const float* d_prs_time_ref, const int* d_active, int mode,
                                         hipStream_t stream) {
    using namespace dabgpu;
    ModeGeom g;
    if (!mode_geometry(mode, g)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ofdm_sync_kernel, dim3((unsigned)n_streams), dim3(256), dabgpu_sync_lds_bytes(g.n_fft), stream,
                       reinterpret_cast<const f2*>(d_prs_syms), stride_samples, n_streams, *cfg, d_states, d_impulse, d_freq,
                       reinterpret_cast<const f2*>(d_tw), reinterpret_cast<const f2*>(d_prs), reinterpret_cast<const f2*>(d_prs_time_ref), d_active, g);
    return hipGetLastError();
}
