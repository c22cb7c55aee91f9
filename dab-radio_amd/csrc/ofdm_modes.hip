// ofdm_modes.hip -- OFDM demodulation of frame-aligned frames for transmission modes II, III and IV (SURVEY 8f row N4;
// geometries of src/ofdm/dab_ofdm_params_ref.cpp:11-60): FFT 512 / 256 / 1024, 384 / 192 / 768 carriers, 76 / 153 / 76 symbols.
// Same pipeline as ofdm_demod.hip -- PLL (apply_pll.cpp:81-117 incl. its scalar tail :115-116 for symbol periods that are
// not a multiple of 4), cyclic-prefix correlation, FFT, DQPSK, frequency de-interleave, soft bits -- written size-generic:
// the transform is a Stockham autosort through LDS with the butterflies and twiddle rule of the mode I contract
// (r1 x 8 x 8 [x 8], r1 = 2 / none / 4), the correlation tree takes one leaf per sample.  Mode I itself runs through the
// register-resident kernel of ofdm_demod.hip; this kernel also accepts mode I, which the tests use to cross-check the two.
// The DAB layer above (FIC_Decoder) exists for mode I only in the reference (fic_decoder.cpp:61-72), so modes II-IV end at
// the soft bits here as they do there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"
#include "iq_sample.h"
#include "ofdm_device.h"

namespace dabgpu {

// BANK (stream bank rounds, ofdm_stream.hip): frame = stream; desc[frame].slot < 0 = nothing to do, else samples [0, split) of
// the frame come from the stream's frame buffer (iq, complex float) and the rest from the caller's block in capture format SRC
// (block + frame * block_stride samples, first at tail_off); the soft bits go to frame slot desc[frame].slot
template <int SRC, bool BANK>
__global__ __launch_bounds__(256)
void ofdm_demod_mode_kernel(int mode, const f2* __restrict__ iq, const float* __restrict__ freq_offset, int8_t* __restrict__ bits,
                            f2* __restrict__ cp_corr, f2* __restrict__ fft_out, const f2* __restrict__ tw,
                            const int* __restrict__ mapper, int n_frames, int sym_per_chunk, int chunks_per_frame,
                            const dabgpu_frame_desc* __restrict__ desc, const uint8_t* __restrict__ block, size_t block_stride)
{
    extern __shared__ __attribute__((aligned(16))) char msm[];
    ModeGeom g;
    mode_geometry(mode, g);
    const int N = g.n_fft;
    f2* Y = reinterpret_cast<f2*>(msm);                 // one PLL-corrected symbol, `period` samples
    f2* W0 = Y + g.period;                              // three transform buffers of N
    float* red = reinterpret_cast<float*>(W0 + 3 * N);  // 2 x 256 reduction leaves
    f2* W[3] = {W0, W0 + N, W0 + 2 * N};

    // the workgroup size is the launcher's choice (128 threads for FFT 512 / 256, else 256)
    const int t = threadIdx.x, NT = blockDim.x;
    const int unit = blockIdx.x;
    const int frame = unit / chunks_per_frame, chunk = unit % chunks_per_frame;
    if (frame >= n_frames) return;
    const int out0 = chunk * sym_per_chunk;
    const int out1 = min(out0 + sym_per_chunk, g.n_sym - 1);
    const bool last_chunk = (out1 == g.n_sym - 1);
    const int sym_end = (last_chunk && fft_out != nullptr) ? g.n_sym : out1;     // inclusive
    const float f = freq_offset ? freq_offset[frame] : 0.0f;
    const f2* fbase = iq + (size_t)frame * g.frame_samples;
    size_t out_frame = (size_t)frame;
    int split = g.frame_samples;
    long long tail_off = 0;
    const uint8_t* tail = nullptr;
    if constexpr (BANK) {
        const dabgpu_frame_desc d = desc[frame];
        if (d.slot < 0) return;
        out_frame = (size_t)d.slot; split = d.split; tail_off = d.tail_off;
        tail = block + (size_t)frame * block_stride * src_sample_bytes<SRC>::value;
    }
    int L = 1;
    while (L < g.n_cp) L <<= 1;
    int prev = 0;                                       // index of the buffer holding the previous symbol's spectrum

    for (int i = out0; i <= sym_end; i++) {
        const float dt0 = (float)(i * g.period) * f;
        const f2* sym = fbase + (size_t)i * g.period;
        for (int n = t; n < g.period; n += NT) {
            f2 v;
            if constexpr (BANK) {
                const int j = i * g.period + n;
                v = (j < split) ? fbase[j] : sample_at<SRC>(tail, tail_off + (j - split));
            } else {
                v = sym[n];
            }
            Y[n] = pll_any(v, n, g.period, f, dt0);
        }
        __syncthreads();
        const bool do_corr = (i < g.n_sym) && (i < out1 || i == g.n_sym - 1);
        if (do_corr) {                                                           // uniform per workgroup
            float pr = 0.0f, pi = 0.0f;
            if (mode == 1) {
                // the mode I contract: leaf t >= 4 = samples 2(t-4), 2(t-4)+1; 64-leaf groups halve with
                // strides 32..1, then (g0+g1)+(g2+g3)
                if (t >= 4) {
                    const int n = 2 * (t - 4);
                    const f2 p0 = conj_mul(Y[N + n], Y[n]), p1 = conj_mul(Y[N + n + 1], Y[n + 1]);
                    pr = p0.x + p1.x; pi = p0.y + p1.y;
                }
                red[t] = pr; red[256 + t] = pi;
                __syncthreads();
                for (int h = 32; h >= 1; h >>= 1) {
                    if ((t & 63) < h) { red[t] += red[t + h]; red[256 + t] += red[256 + t + h]; }
                    __syncthreads();
                }
                if (t == 0) cp_corr[(size_t)frame * g.n_sym + i] = mk2((red[0] + red[64]) + (red[128] + red[192]),
                                                                       (red[256] + red[320]) + (red[384] + red[448]));
            } else {
                // modes II-IV: one leaf per sample, L = power of two >= n_cp (64, 128 or 256), the binary tree red[n] += red[n + h] for
                // h = L/2 .. 1.  The first wavefront does all of it without a barrier: lane n holds leaves n, n + 64, ... (the strides
                // >= 64 are adds inside the lane, in the tree's order), strides 32 .. 1 are lane exchanges; the other wavefronts go on
                if (t < 64) {
                    float lr[4], li[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int n = t + 64 * j;
                        lr[j] = 0.0f; li[j] = 0.0f;
                        if (n < g.n_cp) { const f2 p = conj_mul(Y[N + n], Y[n]); lr[j] = p.x; li[j] = p.y; }
                    }
                    float xr, xi;
                    if (L == 256) { xr = (lr[0] + lr[2]) + (lr[1] + lr[3]); xi = (li[0] + li[2]) + (li[1] + li[3]); }
                    else if (L == 128) { xr = lr[0] + lr[1]; xi = li[0] + li[1]; }
                    else { xr = lr[0]; xi = li[0]; }
                    xr = wave_tree_sum(xr, t); xi = wave_tree_sum(xi, t);
                    if (t == 0) cp_corr[(size_t)frame * g.n_sym + i] = mk2(xr, xi);
                }
            }
        }
        // ---- FFT: Y[cp .. cp + N) -> W[cur]; the two buffers other than `prev` alternate as source and destination ----
        const int wa = (prev + 1) % 3, wb = (prev + 2) % 3;
        const f2* src = Y + g.n_cp;
        int cur_n = N, s = 1, pass = 0, cur = wa;
        const int r1 = (N == 2048 || N == 256) ? 4 : (N == 1024 ? 2 : 8);
        int rem = N;
        while (rem > 1) {
            const int r = (pass == 0) ? r1 : 8;
            const bool last = (rem == r);
            f2* dst = W[(pass & 1) ? wb : wa];
            if (r == 8) stockham_pass<8>(src, dst, N, cur_n, s, last, tw, t, NT);
            else if (r == 4) stockham_pass<4>(src, dst, N, cur_n, s, last, tw, t, NT);
            else stockham_pass<2>(src, dst, N, cur_n, s, last, tw, t, NT);
            __syncthreads();
            src = dst; cur = (pass & 1) ? wb : wa;
            cur_n /= r; s *= r; rem /= r; pass++;
        }
        if (fft_out != nullptr) {
            f2* dst = fft_out + ((size_t)frame * (g.n_sym + 1) + i) * N;
            const lds_cf2 wc = (lds_cf2)W[cur];
            for (int k = t; k < N; k += NT) dst[k] = lds_get(wc, k);
        }
        if (i > out0 && i < g.n_sym) {
            // ---- DQPSK + frequency de-interleave + soft bits (ofdm_demodulator.cpp:842-889) ----
            const int NC = g.n_carriers, M = NC / 2;
            int8_t* o = bits + out_frame * g.frame_bits + (size_t)(i - 1) * g.sym_bits;
            for (int n = t; n < NC; n += NT) {
                const int c = mapper[n];
                const int k = (c < M) ? (c - M) : (c - M + 1);
                const int bin = (N + k) % N;
                const f2 d = conj_mul(lds_get((lds_cf2)W[prev], bin), lds_get((lds_cf2)W[cur], bin));   // (LDS: W is indexed at run time, the casts keep the ds_ path)
                int bx, by;
                soft_bit_pair(d, bx, by);
                o[n] = (int8_t)bx;
                o[n + NC] = (int8_t)by;
            }
        }
        __syncthreads();
        prev = cur;
    }
}

}  // namespace dabgpu

using namespace dabgpu;

// size-generic demodulation launch shared by dabgpu_ofdm_demod_frames_mode and the stream bank (d_desc != nullptr: bank round,
// frame = stream, input split between the bank's frame buffers d_iq and the caller's block d_block in capture format src)
int dabgpu_launch_ofdm_demod_mode(dabgpu_ctx* c, int mode, const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                  float* d_fft, int n_frames, int symbols_per_block, const dabgpu_frame_desc* d_desc, const void* d_block,
                                  size_t block_stride, hipStream_t s) {
    ModeGeom g;
    if (!mode_geometry(mode, g)) { dabgpu_set_error("ofdm_demod_mode: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    int st;
    // modes II-IV without the GUI view run register-resident, one wavefront per run of symbols (ofdm_wave512.hip);
    // DABGPU_MODE_GENERIC=1 keeps them on this file's kernel (the tests cross-check the two)
    if (mode != 1 && !d_fft && !getenv("DABGPU_MODE_GENERIC"))
        return dabgpu_launch_ofdm_demod_wave(c, mode, d_iq, src, d_freq, d_bits, d_cp_corr, n_frames, symbols_per_block, d_desc, d_block, block_stride, s);
    // per-mode carrier mapper on the device, built on first use (get_DAB_mapper_ref, src/ofdm/dab_mapper_ref.cpp:10-51)
    if (!c->d_mode_mapper[mode]) {
        std::vector<int> m((size_t)g.n_carriers);
        if ((st = dabgpu_get_carrier_mapper(mode, m.data()))) return st;
        int* d_m = nullptr;
        if ((st = dabgpu_check_hip(hipMalloc(&d_m, m.size() * sizeof(int)), "hipMalloc(mode mapper)"))) return st;
        if ((st = dabgpu_check_hip(hipMemcpy(d_m, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy(mode mapper)"))) { (void)hipFree(d_m); return st; }
        c->d_mode_mapper[mode] = d_m;
    }
    if (symbols_per_block <= 0 || symbols_per_block > g.n_sym - 1) symbols_per_block = 19;
    const int chunks = (g.n_sym - 1 + symbols_per_block - 1) / symbols_per_block;
    const size_t lds = ((size_t)g.period + 3 * (size_t)g.n_fft) * sizeof(f2) + 512 * sizeof(float);
    // measured (tools/bench_io.py, 2048 frames): FFT 512 / 256 run best with 128 threads (0.94 / 1.01 ms; 256 threads 1.03 / 1.47, 64 threads
    // 1.19 / 1.01), FFT 1024 with 256 (1.93 ms; 128 threads 2.58): fewer idle butterfly lanes against fewer resident wavefronts
    const int n_threads = (mode == 2 || mode == 3) ? 128 : 256;
#define MODE_GO(SRC, BANK)                                                                                                          \
    do {                                                                                                                            \
        if (lds > 48 * 1024 && (st = dabgpu_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(ofdm_demod_mode_kernel<SRC, BANK>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),                   \
                                                      "hipFuncSetAttribute(ofdm_demod_mode_kernel)"))) return st;                  \
        hipLaunchKernelGGL((ofdm_demod_mode_kernel<SRC, BANK>), dim3((unsigned)((size_t)n_frames * chunks)), dim3((unsigned)n_threads), lds, s, mode, \
                           reinterpret_cast<const f2*>(d_iq), d_freq, d_bits, reinterpret_cast<f2*>(d_cp_corr),                    \
                           reinterpret_cast<f2*>(d_fft), reinterpret_cast<const f2*>(c->d_tw), c->d_mode_mapper[mode], n_frames,     \
                           symbols_per_block, chunks, d_desc, static_cast<const uint8_t*>(d_block), block_stride);                  \
    } while (0)
    if (!d_desc) MODE_GO(0, false);
    else if (src == 0) MODE_GO(0, true);
    else if (src == 1) MODE_GO(1, true);
    else if (src == 2) MODE_GO(2, true);
    else MODE_GO(3, true);
#undef MODE_GO
    return dabgpu_check_hip(hipGetLastError(), "ofdm_demod_mode_kernel launch");
}

extern "C" {

int dabgpu_ofdm_demod_frames_mode(dabgpu_ctx* c, int mode, const float* d_iq, size_t n_frames, const float* d_freq, int8_t* d_bits,
                                  float* d_cp_corr, float* d_fft, int symbols_per_block, void* stream) {
    ModeGeom g;
    if (!c || !d_iq || !d_bits) { dabgpu_set_error("ofdm_demod_frames_mode: null ctx/iq/bits"); return DABGPU_ERR_INVALID_ARG; }
    if (!mode_geometry(mode, g)) { dabgpu_set_error("ofdm_demod_frames_mode: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    if (n_frames > (size_t)(1 << 22)) { dabgpu_set_error("ofdm_demod_frames_mode: n_frames too large"); return DABGPU_ERR_INVALID_ARG; }
    if ((uintptr_t)d_iq & 7) { dabgpu_set_error("ofdm_demod_frames_mode: d_iq must be 8-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    if ((uintptr_t)d_bits & 15) { dabgpu_set_error("ofdm_demod_frames_mode: d_bits must be 16-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    int st;
    float* corr = d_cp_corr;
    if (!corr && (st = dabgpu_scratch(c, 23, n_frames * (size_t)g.n_sym * 2 * sizeof(float), (void**)&corr))) return st;
    return dabgpu_launch_ofdm_demod_mode(c, mode, d_iq, 0, d_freq, d_bits, corr, d_fft, (int)n_frames, symbols_per_block, nullptr, nullptr, 0, s);
}

// ---- single-stream, host-buffer forms for the OFDM_Demod mirror class in modes II-IV ----
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)

int dabgpu_ofdm_demod_stream_frame_sync_mode(dabgpu_ctx* c, int mode, const float* h_iq, float freq_coarse, float* h_freq_fine, float beta,
                                             int8_t* h_bits, float* h_total_phase, float* h_fft) {
    ModeGeom g;
    if (!c || !h_iq || !h_bits || !h_freq_fine) { dabgpu_set_error("ofdm_demod_stream_frame_sync_mode: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (!mode_geometry(mode, g)) { dabgpu_set_error("ofdm_demod_stream_frame_sync_mode: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    if (mode == 1) return dabgpu_ofdm_demod_stream_frame_sync(c, h_iq, freq_coarse, h_freq_fine, beta, h_bits, h_total_phase, h_fft, nullptr);
    int st;
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(c);
    const size_t iq_bytes = (size_t)g.frame_samples * 2 * sizeof(float);
    const size_t fft_bytes = (size_t)(g.n_sym + 1) * g.n_fft * 2 * sizeof(float);
    float *d_iq, *d_small, *d_corr, *d_fft = nullptr; int8_t* d_bits;
    if ((st = dabgpu_scratch(c, 1, iq_bytes, (void**)&d_iq))) return st;
    if ((st = dabgpu_scratch(c, 2, (size_t)g.frame_bits, (void**)&d_bits))) return st;
    if ((st = dabgpu_scratch(c, 3, 4 * sizeof(float), (void**)&d_small))) return st;       // [0] net freq, [1] fine, [2] total phase
    if ((st = dabgpu_scratch(c, 4, (size_t)g.n_sym * 2 * sizeof(float), (void**)&d_corr))) return st;
    if (h_fft && (st = dabgpu_scratch(c, 6, fft_bytes, (void**)&d_fft))) return st;
    hipStream_t s = c->stream;
    const float h_small[2] = { freq_coarse + *h_freq_fine, *h_freq_fine };
    CK(hipMemcpyAsync(d_iq, h_iq, iq_bytes, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_small, h_small, sizeof(h_small), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_ofdm_demod_frames_mode(c, mode, d_iq, 1, d_small, d_bits, d_corr, d_fft, 0, s))) return st;
    if ((st = dabgpu_ofdm_phase_update_mode(c, mode, d_corr, 1, beta, d_small + 2, d_small + 1, s))) return st;
    CK(hipMemcpyAsync(h_bits, d_bits, (size_t)g.frame_bits, hipMemcpyDeviceToHost, s));
    float back[2];
    CK(hipMemcpyAsync(back, d_small + 1, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (h_fft) CK(hipMemcpyAsync(h_fft, d_fft, fft_bytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    *h_freq_fine = back[0];
    if (h_total_phase) *h_total_phase = back[1];
    return DABGPU_OK;
}

int dabgpu_ofdm_sync_host_sync_mode(dabgpu_ctx* c, int mode, const float* h_prs_sym, const dabgpu_sync_cfg* cfg, dabgpu_sync_state* h_state,
                                    float* h_impulse, float* h_freq) {
    ModeGeom g;
    if (!c || !h_prs_sym || !cfg || !h_state) { dabgpu_set_error("ofdm_sync_host_sync_mode: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (!mode_geometry(mode, g)) { dabgpu_set_error("ofdm_sync_host_sync_mode: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(c);
    int st;
    const size_t N = (size_t)g.n_fft;
    float *d_sym, *d_imp, *d_frq; dabgpu_sync_state* d_st;
    if ((st = dabgpu_scratch(c, 7, sizeof(float) * 2 * DABGPU_NB_FFT, (void**)&d_sym))) return st;
    if ((st = dabgpu_scratch(c, 8, sizeof(dabgpu_sync_state), (void**)&d_st))) return st;
    if ((st = dabgpu_scratch(c, 9, sizeof(float) * 2 * DABGPU_NB_FFT, (void**)&d_imp))) return st;
    d_frq = d_imp + DABGPU_NB_FFT;
    hipStream_t s = c->stream;
    CK(hipMemcpyAsync(d_sym, h_prs_sym, sizeof(float) * 2 * N, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_st, h_state, sizeof(dabgpu_sync_state), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_ofdm_sync_mode(c, mode, d_sym, 1, N, cfg, d_st, d_imp, d_frq, s))) return st;
    CK(hipMemcpyAsync(h_state, d_st, sizeof(dabgpu_sync_state), hipMemcpyDeviceToHost, s));
    if (h_impulse) CK(hipMemcpyAsync(h_impulse, d_imp, sizeof(float) * N, hipMemcpyDeviceToHost, s));
    if (h_freq) CK(hipMemcpyAsync(h_freq, d_frq, sizeof(float) * N, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return DABGPU_OK;
}
#undef CK

}  // extern "C"
