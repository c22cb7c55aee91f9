// ofdm_wave512.hip -- register-resident OFDM demodulation for transmission modes II (FFT 512), III (FFT 256) and IV (FFT 1024):
// ONE WAVEFRONT per run of symbols, no workgroup barrier anywhere.
//
// The transform contract of the modes (r1 x 8 x 8 x 8 with the twiddles w_n^m = tw2048[m 2048 / n], ofdm_modes.hip /
// oracle dab_fft_n) makes the 512-point transform of mode II identical, operation for operation, to what ONE wavefront of the
// mode I kernel does after that kernel's radix-4 pass (ofdm_demod.hip passes 2-4: 8 points per lane, two 8 x 8 lane<->register
// transposes through a wave-private LDS patch); mode IV adds a radix-2 pass whose two inputs p, p + 512 both belong to lane
// p mod 64, i.e. it runs inside the lane, and then two such 512-point transforms (even / odd outputs).  So here a wavefront
// loads its symbol (8 / 16 samples per lane, 512-byte rows), applies the PLL (apply_pll.cpp incl. the scalar tail of mode II's
// 638-sample period), correlates the cyclic prefix (one leaf per sample, the binary tree of the modes' contract: in-lane adds
// for strides >= 64, lane exchanges below), transforms, multiplies with the previous symbol's six (twelve) active bins it kept
// in registers, and scatters the soft bits to their frequency-de-interleaved positions in a wave-private LDS row that leaves
// as 16-byte stores.  Mode III (256 = 4 x 8 x 8) does its radix-4 pass inside the lane too and the 64-point sub-transforms on half
// of the lanes (fft256_wave).  Four independent wavefronts per workgroup; the size-generic LDS-Stockham kernel of ofdm_modes.hip
// (1 TB/s, barrier-bound) remains for the GUI views (fft_out) and as the cross-check of this one in the tests.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"
#include "iq_sample.h"
#include "ofdm_device.h"

// minimum waves per SIMD asked of the compiler for mode IV: 1 = no cap (140 VGPRs, three waves per SIMD, 0.45 ms per 2048 frames);
// 4 caps at 128 registers, spills 10 and measured 0.51 ms (the register-table version before: 176 VGPRs, two waves, 0.48 ms)
#ifndef W512_M4_WAVES
#define W512_M4_WAVES 1
#endif

namespace dabgpu {

template <int MODE> struct W512Geom;
template <> struct W512Geom<2> { static constexpr int N = 512, CP = 126, PERIOD = 638, NSYM = 76, NC = 384, NULLP = 664; };
template <> struct W512Geom<4> { static constexpr int N = 1024, CP = 252, PERIOD = 1276, NSYM = 76, NC = 768, NULLP = 1328; };
template <> struct W512Geom<3> { static constexpr int N = 256, CP = 63, PERIOD = 319, NSYM = 153, NC = 192, NULLP = 345; };

// mode III, 256 = 4 x 8 x 8: the radix-4 pass runs inside the lane (inputs lane + 64 j), the two radix-8 passes of the four 64-point
// sub-transforms on 32 lanes (8 points each) around two LDS transposes in Stockham order.  x[j] = input lane + 64 j  ->  lanes 0..31:
// a[k] = bin lane + 32 k
__device__ __forceinline__ void fft256_wave(const f2 (&x)[4], f2 (&a)[8], f2* patch, const f2 (&w1)[3], const f2 (&w2)[7], int lane) {
    f2 b0, b1, b2, b3;
    dft4(x[0], x[1], x[2], x[3], b0, b1, b2, b3);
    patch[lane] = b0;                                            // pass-1 output 4 p + k as [k][p]
    patch[64 + lane] = cmul(b1, w1[0]);
    patch[128 + lane] = cmul(b2, w1[1]);
    patch[192 + lane] = cmul(b3, w1[2]);
    wave_lds_fence();
    const int q = (lane >> 3) & 3, p2 = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[64 * q + p2 + 8 * j];
    wave_lds_fence();
    dft8(a);
    if (lane < 32) {
        f2* pb = patch + 256;                                    // pass-2 output q + 4 (8 p2 + k)
        pb[q + 32 * p2] = a[0];
#pragma unroll
        for (int k = 1; k < 8; k++) pb[q + 32 * p2 + 4 * k] = cmul(a[k], w2[k - 1]);
    }
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[256 + (lane & 31) + 32 * j];
    wave_lds_fence();
    dft8(a);
}

// the 512-point transform of one wavefront: a[j] = input lane + 64 j  ->  a[k] = bin (lane >> 3) + 8 (lane & 7) + 64 k
// LW: the twiddles come from the workgroup's LDS tables (w2p[64 (k - 1)], w3p[8 (k - 1)]; mode IV, whose sixteen samples per lane
// leave no registers for them) instead of registers -- the same table entries
template <bool LW>
__device__ __forceinline__ void fft512_wave(f2 (&a)[8], f2* patch, const f2 (&w2)[7], const f2 (&w3)[7], const f2* w2p, const f2* w3p, int lane) {
    const int la = lane & 7, lb = lane >> 3;
    const int ta_w = lane, ta_r = la + 72 * lb, tb_w = la + 72 * lb, tb_r = 9 * la + 72 * lb;
    dft8(a);
    f2 t2[7], t3[7];
    if constexpr (LW) {
#pragma unroll
        for (int k = 0; k < 7; k++) t2[k] = w2p[64 * k];
    }
    patch[ta_w] = a[0];
#pragma unroll
    for (int k = 1; k < 8; k++) patch[ta_w + 72 * k] = cmul(a[k], LW ? t2[k - 1] : w2[k - 1]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[ta_r + 8 * j];
    wave_lds_fence();
    dft8(a);
    if constexpr (LW) {
#pragma unroll
        for (int k = 0; k < 7; k++) t3[k] = w3p[8 * k];
    }
    patch[tb_w] = a[0];
#pragma unroll
    for (int k = 1; k < 8; k++) patch[tb_w + 9 * k] = cmul(a[k], LW ? t3[k - 1] : w3[k - 1]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = patch[tb_r + j];
    wave_lds_fence();
    dft8(a);
}

template <int MODE, int SRC, bool BANK>
__global__ __launch_bounds__(256, MODE == 4 ? W512_M4_WAVES : 1)
void ofdm_demod_wave_kernel(const f2* __restrict__ iq, const float* __restrict__ freq_offset, int8_t* __restrict__ bits,
                            f2* __restrict__ cp_corr, const f2* __restrict__ tw, const int* __restrict__ inv_map, int n_frames,
                            int sym_per_chunk, int chunks_per_frame, const dabgpu_frame_desc* __restrict__ desc,
                            const uint8_t* __restrict__ block, size_t block_stride)
{
    using G = W512Geom<MODE>;
    constexpr int N = G::N, CP = G::CP, PERIOD = G::PERIOD, NSYM = G::NSYM, NC = G::NC;
    constexpr int P = N / 64;                       // samples per lane
    constexpr int Q = (N == 1024) ? 2 : 1;          // 512-point sub-transforms (mode IV: even / odd outputs of the radix-2 pass)
    constexpr int STRIDE = N / 8;                   // bin distance between the registers of a lane after the last pass
    constexpr int SYM_BITS = 2 * NC, FRAME_BITS = (NSYM - 1) * SYM_BITS, FRAME_SAMPLES = NSYM * PERIOD + G::NULLP;
    constexpr int LEAVES = (CP + 63) / 64;          // correlation leaves per lane: the tree has 64 LEAVES leaves (128 / 256)
    constexpr int TAIL0 = N - CP;                   // body index of the sample that pairs with cyclic-prefix sample 0

    __shared__ __attribute__((aligned(16))) f2 patches[4][WAVE_PATCH];
    __shared__ __attribute__((aligned(16))) int8_t obufs[4][SYM_BITS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f2* patch = patches[wave];
    int8_t* obuf = obufs[wave];
    // mode IV keeps its three twiddle sets in LDS tables of the workgroup (filled here, by all four waves, before any of them can
    // leave: the only workgroup barrier of the kernel)
    constexpr bool LW = (MODE == 4);
    __shared__ f2 tw_lds[LW ? (8 * 64 + 7 * 64 + 7 * 8) : 1];
    if constexpr (LW) {
        for (int e = (int)threadIdx.x; e < 8 * 64; e += 256) tw_lds[e] = tw[2 * ((e & 63) + 64 * (e >> 6))];                 // w0 [j][lane]
        for (int e = (int)threadIdx.x; e < 7 * 64; e += 256) tw_lds[8 * 64 + e] = tw[4 * (e & 63) * ((e >> 6) + 1)];           // w2 [k - 1][lane]
        for (int e = (int)threadIdx.x; e < 7 * 8; e += 256) tw_lds[15 * 64 + e] = tw[32 * (e & 7) * ((e >> 3) + 1)];           // w3 [k - 1][lane & 7]
        __syncthreads();
    }

    const int unit = (int)blockIdx.x * 4 + wave;
    const int frame = unit / chunks_per_frame, chunk = unit % chunks_per_frame;
    if (frame >= n_frames) return;
    const f2* fbase = iq + (size_t)frame * FRAME_SAMPLES;
    size_t out_frame = (size_t)frame;
    int split = FRAME_SAMPLES;
    long long tail_off = 0;
    const uint8_t* tail = nullptr;
    if constexpr (BANK) {
        const dabgpu_frame_desc d = desc[frame];
        if (d.slot < 0) return;
        out_frame = (size_t)d.slot; split = d.split; tail_off = d.tail_off;
        tail = block + (size_t)frame * block_stride * src_sample_bytes<SRC>::value;
    }
    auto fetch = [&](int j) -> f2 {                 // sample j of the frame
        if constexpr (BANK) return (j < split) ? fbase[j] : sample_at<SRC>(tail, tail_off + (j - split));
        else return fbase[j];
    };

    const int out0 = chunk * sym_per_chunk;
    const int out1 = min(out0 + sym_per_chunk, NSYM - 1);
    const float f = freq_offset ? freq_offset[frame] : 0.0f;

    // twiddles resident in registers: pass r1 (mode IV) w_1024^{p}, p = lane + 64 j; then w_512^{lane k}, w_64^{(lane & 7) k}
    // (mode III: pass 1 w_256^{lane k}, pass 2 w_64^{(lane & 7) k})
    // (mode IV: sixteen samples per lane and twelve kept bins leave no registers for 44 twiddle registers -- 176 VGPRs, two waves per
    // SIMD; its three sets sit in LDS tables of the workgroup instead: 8.4 KB, read just before use)
    f2 w0[8], w2[7], w3[7];
    if constexpr (LW) {
#pragma unroll
        for (int j = 0; j < 8; j++) w0[j] = mk2(1.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < 7; k++) { w2[k] = mk2(1.0f, 0.0f); w3[k] = mk2(1.0f, 0.0f); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) w0[j] = (Q == 2) ? tw[2 * (lane + 64 * j)] : ((MODE == 3 && j >= 1 && j < 4) ? tw[8 * lane * j] : mk2(1.0f, 0.0f));
#pragma unroll
        for (int k = 1; k < 8; k++) { w2[k - 1] = tw[4 * lane * k]; w3[k - 1] = tw[32 * (lane & 7) * k]; }
    }
    const f2* const w0p = tw_lds + lane;
    const f2* const w2p = tw_lds + (LW ? 8 * 64 : 0) + lane;
    const f2* const w3p = tw_lds + (LW ? 15 * 64 : 0) + (lane & 7);

    // soft-bit positions of the lane's active bins.  Sub-transform q leaves bin q + Q (Ks + 64 k) in register k, Ks = (lane >> 3) +
    // 8 (lane & 7); carriers -NC/2 .. -1, 1 .. NC/2 are bins N - NC/2 .. N - 1 and 1 .. NC/2 = registers 5, 6, 7 and 0, 1, 2 (+ register 3
    // for bin NC/2 itself, which takes the place of the DC bin in the lane that holds both); carrier index c = bin - (N - NC/2) or
    // bin + NC/2 - 1
    // (mode III: lanes 0..31 hold bins lane + 32 k, the others none)
    const int Ks = (MODE == 3) ? (lane & 31) : ((lane >> 3) + 8 * (lane & 7));
    const bool has_bins = (MODE != 3) || lane < 32;
    int pos[Q][6];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int b0 = q + Q * Ks;                  // bin of register 0
        pos[q][0] = inv_map[(b0 == 0) ? (NC - 1) : (b0 + NC / 2 - 1)];
        pos[q][1] = inv_map[b0 + STRIDE + NC / 2 - 1];
        pos[q][2] = inv_map[b0 + 2 * STRIDE + NC / 2 - 1];
        pos[q][3] = inv_map[b0 + 5 * STRIDE - (N - NC / 2)];
        pos[q][4] = inv_map[b0 + 6 * STRIDE - (N - NC / 2)];
        pos[q][5] = inv_map[b0 + 7 * STRIDE - (N - NC / 2)];
    }
    f2 prev[Q][6];
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
        for (int k = 0; k < 6; k++) prev[q][k] = mk2(0.0f, 0.0f);

    for (int i = out0; i <= out1; i++) {
        const float dt0 = (float)(i * PERIOD) * f;                       // ofdm_demodulator.cpp:675-676
        const int s0 = i * PERIOD;
        // ---- load + PLL: lane holds body samples lane + 64 j ----
        f2 x[P];
#pragma unroll
        for (int j = 0; j < P; j++) { const int n = CP + lane + 64 * j; x[j] = pll_any(fetch(s0 + n), n, PERIOD, f, dt0); }

        // ---- cyclic prefix correlation: leaf n = prefix sample n x conj... (CalculateCyclicPhaseError :768-777), n = lane + 64 m ----
        const bool do_corr = (i < out1 || i == NSYM - 1);
        if (do_corr) {                                                   // wave-uniform
            constexpr int SH = TAIL0 & 63, RB = TAIL0 >> 6;              // the partner of prefix sample n is body sample TAIL0 + n
            const int src = (lane + SH) & 63;
            const bool wrap = lane + SH >= 64;
            float lr[LEAVES], li[LEAVES];
#pragma unroll
            for (int m = 0; m < LEAVES; m++) {
                const int n = lane + 64 * m;
                // body sample TAIL0 + n lives in lane `src`, register RB + m (+ 1 when the lane index wrapped)
                const float ax = __shfl(x[RB + m].x, src), ay = __shfl(x[RB + m].y, src);
                float bx = 0.0f, by = 0.0f;
                if (RB + m + 1 < P) { bx = __shfl(x[(RB + m + 1 < P) ? (RB + m + 1) : 0].x, src); by = __shfl(x[(RB + m + 1 < P) ? (RB + m + 1) : 0].y, src); }
                const f2 tl = wrap ? mk2(bx, by) : mk2(ax, ay);
                lr[m] = 0.0f; li[m] = 0.0f;
                if (n < CP) {
                    const f2 hd = pll_any(fetch(s0 + n), n, PERIOD, f, dt0);
                    const f2 p = conj_mul(tl, hd);
                    lr[m] = p.x; li[m] = p.y;
                }
            }
            float xr, xi;
            if constexpr (LEAVES == 4) { xr = (lr[0] + lr[2]) + (lr[1] + lr[3]); xi = (li[0] + li[2]) + (li[1] + li[3]); }
            else if constexpr (LEAVES == 2) { xr = lr[0] + lr[1]; xi = li[0] + li[1]; }
            else { xr = lr[0]; xi = li[0]; }
            const float xs = wave_tree_sum_pair(xr, xi);                 // lane 0: the sum of xr, lane 32: the sum of xi (same additions)
            if ((lane & 31) == 0) reinterpret_cast<float*>(cp_corr + (size_t)frame * NSYM + i)[lane >> 5] = xs;
        }

        // ---- transform(s) + DQPSK + soft bits ----
        const bool emit = (i > out0);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            f2 a[8];
            if constexpr (MODE == 3) {
                const f2 w1[3] = {w0[1], w0[2], w0[3]};
                fft256_wave(x, a, patch, w1, w3, lane);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if constexpr (Q == 1) a[j] = x[j];
                    else a[j] = (q == 0) ? (x[j] + x[j + 8]) : cmul(x[j] - x[j + 8], LW ? w0p[64 * j] : w0[j]);     // radix-2 pass, inside the lane
                }
                fft512_wave<LW>(a, patch, w2, w3, w2p, w3p, lane);
            }
            f2 cur[6];
            cur[0] = (q == 0 && Ks == 0) ? a[3] : a[0];
            cur[1] = a[1]; cur[2] = a[2]; cur[3] = a[5]; cur[4] = a[6]; cur[5] = a[7];
            if (emit && has_bins) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    int bx, by;
                    soft_bit_pair(conj_mul(prev[q][k], cur[k]), bx, by);
                    obuf[pos[q][k]] = (int8_t)bx;
                    obuf[pos[q][k] + NC] = (int8_t)by;
                }
            }
#pragma unroll
            for (int k = 0; k < 6; k++) prev[q][k] = cur[k];
        }
        wave_lds_fence();
        if (emit) {
            uint4* dst = reinterpret_cast<uint4*>(bits + out_frame * FRAME_BITS + (size_t)(i - 1) * SYM_BITS);
#pragma unroll
            for (int c = lane; c < SYM_BITS / 16; c += 64) dst[c] = reinterpret_cast<const uint4*>(obuf)[c];
        }
        wave_lds_fence();
    }
}


// ---- mode III, two symbols per wavefront ----
// 256 = 4 x 8 x 8 leaves bins on 32 lanes only: in ofdm_demod_wave_kernel<3> the two radix-8 passes, their twiddle products and the
// demapper run with half of the lanes idle.  Here a wavefront walks its run of symbols in PAIRS: the PLL and the in-lane radix-4
// pass take both symbols on all 64 lanes (4 samples per lane and symbol), then lanes 0..31 carry the 64-point sub-transforms and
// the demapper of symbol i, lanes 32..63 those of symbol i + 1.  DQPSK needs the bins of the symbol before: lanes 32..63 take
// symbol i's from the lower half of this pair, lanes 0..31 symbol i - 1's from the upper half of the previous pair -- one
// v_permlane32_swap + select per kept float.  Operation for operation the arithmetic of fft256_wave / the generic path (same
// tables, same butterflies, same trees), so the results are bit-identical; an odd last symbol runs with the upper half masked.
template <int SRC, bool BANK>
__global__ __launch_bounds__(256)
void ofdm_demod_wave3_kernel(const f2* __restrict__ iq, const float* __restrict__ freq_offset, int8_t* __restrict__ bits,
                             f2* __restrict__ cp_corr, const f2* __restrict__ tw, const int* __restrict__ inv_map, int n_frames,
                             int sym_per_chunk, int chunks_per_frame, const dabgpu_frame_desc* __restrict__ desc,
                             const uint8_t* __restrict__ block, size_t block_stride)
{
    using G = W512Geom<3>;
    constexpr int N = G::N, CP = G::CP, PERIOD = G::PERIOD, NSYM = G::NSYM, NC = G::NC;
    constexpr int SYM_BITS = 2 * NC, FRAME_BITS = (NSYM - 1) * SYM_BITS, FRAME_SAMPLES = NSYM * PERIOD + G::NULLP;
    constexpr int TAIL0 = N - CP;                   // body index of the sample that pairs with cyclic-prefix sample 0
    constexpr int STRIDE = N / 8;

    __shared__ __attribute__((aligned(16))) f2 patches[4][4 * 256];        // per wave: pass-1 outputs of both symbols, pass-2 outputs of both
    __shared__ __attribute__((aligned(16))) int8_t obufs[4][2][SYM_BITS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f2* patch = patches[wave];
    const int h = lane >> 5, l5 = lane & 31;         // half (symbol of the pair) and lane inside the half
    int8_t* obuf = obufs[wave][h];

    const int unit = (int)blockIdx.x * 4 + wave;
    const int frame = unit / chunks_per_frame, chunk = unit % chunks_per_frame;
    if (frame >= n_frames) return;
    const f2* fbase = iq + (size_t)frame * FRAME_SAMPLES;
    size_t out_frame = (size_t)frame;
    int split = FRAME_SAMPLES;
    long long tail_off = 0;
    const uint8_t* tail = nullptr;
    if constexpr (BANK) {
        const dabgpu_frame_desc d = desc[frame];
        if (d.slot < 0) return;
        out_frame = (size_t)d.slot; split = d.split; tail_off = d.tail_off;
        tail = block + (size_t)frame * block_stride * src_sample_bytes<SRC>::value;
    }
    auto fetch = [&](int j) -> f2 {                 // sample j of the frame
        if constexpr (BANK) return (j < split) ? fbase[j] : sample_at<SRC>(tail, tail_off + (j - split));
        else return fbase[j];
    };
    const int out0 = chunk * sym_per_chunk;
    const int out1 = min(out0 + sym_per_chunk, NSYM - 1);
    const float f = freq_offset ? freq_offset[frame] : 0.0f;

    f2 w1[3], w3[7];                                 // pass 1 w_256^{lane k}, pass 2 w_64^{(lane & 7) k}
#pragma unroll
    for (int k = 1; k < 4; k++) w1[k - 1] = tw[8 * lane * k];
#pragma unroll
    for (int k = 1; k < 8; k++) w3[k - 1] = tw[32 * (lane & 7) * k];

    const int Ks = l5;                               // register k of lane (h, l5) ends with bin l5 + 32 k of symbol i + h
    int pos[6];
    pos[0] = inv_map[(Ks == 0) ? (NC - 1) : (Ks + NC / 2 - 1)];
    pos[1] = inv_map[Ks + STRIDE + NC / 2 - 1];
    pos[2] = inv_map[Ks + 2 * STRIDE + NC / 2 - 1];
    pos[3] = inv_map[Ks + 5 * STRIDE - (N - NC / 2)];
    pos[4] = inv_map[Ks + 6 * STRIDE - (N - NC / 2)];
    pos[5] = inv_map[Ks + 7 * STRIDE - (N - NC / 2)];
    f2 last[6];                                      // the kept bins this lane produced in the previous pair
#pragma unroll
    for (int k = 0; k < 6; k++) last[k] = mk2(0.0f, 0.0f);

    constexpr int SH = TAIL0 & 63, RB = TAIL0 >> 6;  // the partner of prefix sample n is body sample TAIL0 + n
    const int csrc = (lane + SH) & 63;
    const bool cwrap = lane + SH >= 64;

    for (int i = out0; i <= out1; i += 2) {
        const bool second = (i + 1 <= out1);         // (uniform) the pair has its second symbol
        const int sym[2] = {i, second ? i + 1 : i};  // an odd last symbol is loaded twice, its copy is never used
        // ---- both symbols on all lanes: PLL, correlation leaf, radix 4 ----
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int si = sym[u];
            const float dt0 = (float)(si * PERIOD) * f;                    // ofdm_demodulator.cpp:675-676
            const int s0 = si * PERIOD;
            f2 x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int n = CP + lane + 64 * j; x[j] = pll_any(fetch(s0 + n), n, PERIOD, f, dt0); }
            const bool do_corr = (u == 0 || second) && (si < out1 || si == NSYM - 1);       // (uniform)
            if (do_corr) {
                const float ax = __shfl(x[RB].x, csrc), ay = __shfl(x[RB].y, csrc);
                float bx = 0.0f, by = 0.0f;
                if (RB + 1 < 4) { bx = __shfl(x[(RB + 1 < 4) ? (RB + 1) : 0].x, csrc); by = __shfl(x[(RB + 1 < 4) ? (RB + 1) : 0].y, csrc); }
                const f2 tl = cwrap ? mk2(bx, by) : mk2(ax, ay);
                float lr = 0.0f, li = 0.0f;
                if (lane < CP) {
                    const f2 hd = pll_any(fetch(s0 + lane), lane, PERIOD, f, dt0);
                    const f2 p = conj_mul(tl, hd);
                    lr = p.x; li = p.y;
                }
                const float xs = wave_tree_sum_pair(lr, li);
                if ((lane & 31) == 0) reinterpret_cast<float*>(cp_corr + (size_t)frame * NSYM + si)[lane >> 5] = xs;
            }
            f2 b0, b1, b2, b3;
            dft4(x[0], x[1], x[2], x[3], b0, b1, b2, b3);
            f2* pa = patch + 256 * u;                                      // pass-1 output 4 p + k of symbol u as [k][p]
            pa[lane] = b0;
            pa[64 + lane] = cmul(b1, w1[0]);
            pa[128 + lane] = cmul(b2, w1[1]);
            pa[192 + lane] = cmul(b3, w1[2]);
        }
        wave_lds_fence();
        // ---- each half its symbol: two radix-8 passes of the four 64-point sub-transforms ----
        f2 a[8];
        const int q = (l5 >> 3) & 3, p2 = l5 & 7;
        {
            const f2* pa = patch + 256 * h;
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = pa[64 * q + p2 + 8 * j];
        }
        wave_lds_fence();
        dft8(a);
        {
            f2* pb = patch + 512 + 256 * h;                                // pass-2 output q + 4 (8 p2 + k)
            pb[q + 32 * p2] = a[0];
#pragma unroll
            for (int k = 1; k < 8; k++) pb[q + 32 * p2 + 4 * k] = cmul(a[k], w3[k - 1]);
        }
        wave_lds_fence();
        {
            const f2* pb = patch + 512 + 256 * h;
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = pb[l5 + 32 * j];
        }
        wave_lds_fence();
        dft8(a);
        f2 cur[6];
        cur[0] = (Ks == 0) ? a[3] : a[0];
        cur[1] = a[1]; cur[2] = a[2]; cur[3] = a[5]; cur[4] = a[6]; cur[5] = a[7];
        // ---- DQPSK against the symbol before: lower half <- upper half of the previous pair, upper half <- lower half of this pair ----
        const bool emit = h ? second : (i > out0);                          // symbol i + h produces row i + h - 1
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(last[k].x), __float_as_uint(cur[k].x), false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(last[k].y), __float_as_uint(cur[k].y), false, false);
            // swap(a = last, b = cur): a' = [a.lo | b.lo], b' = [a.hi | b.hi] -> lower lanes want a.hi = b'.lo, upper lanes b.lo = a'.hi
            const f2 pv = mk2(__uint_as_float(h ? sx[0] : sx[1]), __uint_as_float(h ? sy[0] : sy[1]));
            if (emit) {
                int bx, by;
                soft_bit_pair(conj_mul(pv, cur[k]), bx, by);
                obuf[pos[k]] = (int8_t)bx;
                obuf[pos[k] + NC] = (int8_t)by;
            }
            last[k] = cur[k];
        }
        wave_lds_fence();
        if (emit && l5 < SYM_BITS / 16) {
            uint4* dst = reinterpret_cast<uint4*>(bits + out_frame * FRAME_BITS + (size_t)(i + h - 1) * SYM_BITS);
            dst[l5] = reinterpret_cast<const uint4*>(obuf)[l5];
        }
        wave_lds_fence();
    }
}

}  // namespace dabgpu

using namespace dabgpu;

// modes II, III and IV without the GUI views; frame = stream when d_desc != nullptr (stream bank rounds)
int dabgpu_launch_ofdm_demod_wave(dabgpu_ctx* c, int mode, const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                  int n_frames, int symbols_per_block, const dabgpu_frame_desc* d_desc, const void* d_block,
                                  size_t block_stride, hipStream_t s) {
    ModeGeom g;
    if ((mode != 2 && mode != 3 && mode != 4) || !mode_geometry(mode, g)) { dabgpu_set_error("ofdm_demod_wave: mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    int st;
    // inverse of the frequency interleaver on the device, built on first use: carrier c carries soft bit inv[c] (get_DAB_mapper_ref)
    if (!c->d_mode_inv_map[mode]) {
        std::vector<int> m((size_t)g.n_carriers), inv((size_t)g.n_carriers);
        if ((st = dabgpu_get_carrier_mapper(mode, m.data()))) return st;
        for (int n = 0; n < g.n_carriers; n++) inv[(size_t)m[(size_t)n]] = n;
        // (into a local first: a failed copy must not leave a non-null pointer to uninitialised indices on the context)
        int* d_inv = nullptr;
        if ((st = dabgpu_check_hip(hipMalloc(&d_inv, inv.size() * sizeof(int)), "hipMalloc(mode inverse mapper)"))) return st;
        if ((st = dabgpu_check_hip(hipMemcpy(d_inv, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy(mode inverse mapper)"))) { (void)hipFree(d_inv); return st; }
        c->d_mode_inv_map[mode] = d_inv;
    }
    if (symbols_per_block <= 0 || symbols_per_block > g.n_sym - 1) symbols_per_block = 19;
    const int chunks = (g.n_sym - 1 + symbols_per_block - 1) / symbols_per_block;
    const size_t units = (size_t)n_frames * chunks;
    const dim3 grid((unsigned)((units + 3) / 4));
#define WAVE_GO(MODE, SRC, BANK)                                                                                                   \
    hipLaunchKernelGGL((ofdm_demod_wave_kernel<MODE, SRC, BANK>), grid, dim3(256), 0, s, reinterpret_cast<const f2*>(d_iq), d_freq, d_bits, \
                       reinterpret_cast<f2*>(d_cp_corr), reinterpret_cast<const f2*>(c->d_tw), c->d_mode_inv_map[mode], n_frames,    \
                       symbols_per_block, chunks, d_desc, static_cast<const uint8_t*>(d_block), block_stride)
#define WAVE_MODE(MODE)                                          \
    do {                                                         \
        if (!d_desc) WAVE_GO(MODE, 0, false);                    \
        else if (src == 0) WAVE_GO(MODE, 0, true);               \
        else if (src == 1) WAVE_GO(MODE, 1, true);               \
        else if (src == 2) WAVE_GO(MODE, 2, true);               \
        else WAVE_GO(MODE, 3, true);                             \
    } while (0)
#define WAVE3_GO(SRC, BANK)                                                                                                          \
    hipLaunchKernelGGL((ofdm_demod_wave3_kernel<SRC, BANK>), grid, dim3(256), 0, s, reinterpret_cast<const f2*>(d_iq), d_freq, d_bits,    \
                       reinterpret_cast<f2*>(d_cp_corr), reinterpret_cast<const f2*>(c->d_tw), c->d_mode_inv_map[mode], n_frames,     \
                       symbols_per_block, chunks, d_desc, static_cast<const uint8_t*>(d_block), block_stride)
    if (mode == 2) WAVE_MODE(2);
    else if (mode == 4) WAVE_MODE(4);
    else if (getenv("DABGPU_MODE3_SINGLE")) WAVE_MODE(3);          // development: one symbol per wavefront (the cross-check of the tests)
    else if (!d_desc) WAVE3_GO(0, false);
    else if (src == 0) WAVE3_GO(0, true);
    else if (src == 1) WAVE3_GO(1, true);
    else if (src == 2) WAVE3_GO(2, true);
    else WAVE3_GO(3, true);
#undef WAVE3_GO
#undef WAVE_MODE
#undef WAVE_GO
    return dabgpu_check_hip(hipGetLastError(), "ofdm_demod_wave_kernel launch");
}
