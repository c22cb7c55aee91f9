// viterbi.hip -- gfx950 kernel for the channel-decode half of the hot path.
//
// One wavefront decodes one punctured K=7 rate-1/4 codeword (FIB group or sub-channel logical frame):
//   lane j            = trellis state j (64 states = 64 lanes)
//   branch metric     = 508 - v_dot4_i32_i8(sign pattern of the lane, 4 packed soft symbols)
//   add-compare-select: predecessors j>>1 and (j>>1)+32 fetched with two ds_bpermute; u16 metrics with the
//                       reference's renormalisation rule (dab_viterbi_decoder.cpp:31-41)
//   decision word     = the 64-bit compare mask (v_cmp writes it straight into an SGPR pair), bit j = state j
//   de-puncturing     = index arithmetic on the kept-count tables (dab_viterbi_decoder.cpp:131-181), never materialised
//   time de-interleave= index arithmetic on a ring of past CIFs (cif_deinterleaver.cpp:36-71), never materialised
//   chain-back        = scalar pointer chase over decision words parked in a per-wave HBM/L2 scratch ring,
//                       MSB-first bytes, XOR with the energy-dispersal PRBS (additive_scrambler.h:16-35),
//                       CRC16 of each FIB (fic_decoder.cpp:103-116) by three lanes
// Restates williamyang98/ViterbiDecoderCpp's scalar core (vendor/viterbi_decoder, empty submodule in the reference
// snapshot) behind DAB_Viterbi_Decoder's call sites; tie_rule 0 = scalar core, 1 = SIMD cores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabgpu_internal.h"

namespace dabgpu {

constexpr uint32_t V_NONSTART = 5080u;
constexpr uint32_t V_RENORM = 60455u;
constexpr int PRBS_PERIOD = 511;

__device__ __forceinline__ int parity7(unsigned v) { return __builtin_popcount(v & 0x7Fu) & 1; }

// kept-count (low nibble) and running prefix (high bits) of puncture vector PI_pi, 4-bit group g (0..7):
// the n-th extra kept bit of a 32-bit period goes to group bitrev3(n mod 8) (ETSI EN 300 401 table 13;
// same data as src/dab/constants/puncture_codes.h:42-67).  PI_X (tail) == PI_8 restricted to 6 groups.
__device__ __forceinline__ int pi_count(int pi, int g) {
    const int ord = ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);
    return 1 + ((pi > ord) ? (((pi - 1 - ord) >> 3) + 1) : 0);
}

__global__ __launch_bounds__(64)
void viterbi_kernel(const dabgpu_cw_desc* __restrict__ descs, int n_cw, uint64_t* __restrict__ dec_scratch,
                    size_t scratch_words_per_wave, dabgpu_cw_result* __restrict__ results, int tie_rule)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vsm[];
    uint16_t* pi_tab = reinterpret_cast<uint16_t*>(vsm);            // [25][8]: count | prefix << 8
    unsigned char* prbs = vsm + 25 * 8 * 2;                         // [512]
    unsigned char* obytes = prbs + 512;                             // decoded bytes of the current codeword

    const int lane = threadIdx.x;
    // ---- tables ----
    for (int e = lane; e < 25 * 8; e += 64) {
        const int pi = e >> 3, g = e & 7;
        int pre = 0;
        for (int q = 0; q < g; q++) pre += pi_count(pi, q);
        pi_tab[e] = (uint16_t)(pi_count(pi, g) | (pre << 8));
    }
    if (lane == 0) {                                                // additive_scrambler.h:16-35, syncword 0xFFFF
        unsigned reg = 0xFFFFu;
        for (int k = 0; k < PRBS_PERIOD; k++) {
            unsigned b = 0;
            for (int i = 0; i < 8; i++) {
                const unsigned v = ((reg >> 8) ^ (reg >> 4)) & 1u;
                b |= v << (7 - i);
                reg = ((reg << 1) | v) & 0xFFFFu;
            }
            prbs[k] = (unsigned char)b;
        }
    }
    __syncthreads();

    // ---- per-lane trellis constants ----
    // new state j = lane: butterfly s = j>>1, input bit b = j&1.  Branch table sign for symbol r of butterfly s is
    // +127 if parity((2s) & G_r) (ViterbiBranchTable), error = sum_r |branch_r - y_r| = 508 - sum_r sigma_r y_r,
    // and the other transition of the butterfly costs 1016 - error: fold that into the sign for odd lanes.
    const int s_idx = lane >> 1, b_in = lane & 1;
    const unsigned G[4] = {109u, 79u, 83u, 109u};                   // dab_viterbi_decoder.cpp:25
    int sig = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int sg = parity7((unsigned)(2 * s_idx) & G[r]) ? 1 : -1;
        if (b_in) sg = -sg;
        sig |= (sg & 0xFF) << (8 * r);
    }
    const int bp_lo = 4 * s_idx, bp_hi = 4 * (s_idx + 32);

    uint64_t* my_dec = dec_scratch + (size_t)blockIdx.x * scratch_words_per_wave;

    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const dabgpu_cw_desc D = descs[cw];
        const int n_steps = (int)D.n_steps;
        // segment boundaries in trellis steps; the 6 tail steps use PI_8 == PI_X
        int seg_end[5], seg_pi[5], seg_in0[5];
        {
            int st = 0, in0 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                seg_pi[k] = (int)D.seg_pi[k];
                seg_in0[k] = in0;
                st += (int)D.seg_steps[k];
                in0 += ((int)D.seg_steps[k] >> 3) * (8 + seg_pi[k]);
                seg_end[k] = st;
            }
            seg_pi[4] = 8; seg_in0[4] = in0; seg_end[4] = st + 6;
        }

        uint32_t metric = (lane == (int)(D.start_state & 63u)) ? 0u : V_NONSTART;
        uint64_t renorm_total = 0;
        uint32_t vlo = 0, vhi = 0;

        for (int t0 = 0; t0 < n_steps; t0 += 16) {
            // ---- fetch + de-puncture + time de-interleave: lane L owns mother symbol 4*t0 + L ----
            int ypk;
            {
                const int M = 4 * t0 + lane;
                const int step = M >> 2, r = M & 3;
                int k = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) k += (step >= seg_end[q]) ? 1 : 0;
                int sstart = 0, pi = seg_pi[0], in0 = seg_in0[0];
#pragma unroll
                for (int q = 1; q < 5; q++) if (k == q) { sstart = seg_end[q - 1]; pi = seg_pi[q]; in0 = seg_in0[q]; }
                const int sis = step - sstart;
                const uint16_t e = pi_tab[pi * 8 + (sis & 7)];
                const int cnt = e & 0xFF, pre = e >> 8;
                const int idx = in0 + (sis >> 3) * (8 + pi) + pre + r;
                int y = 0;
                if (step < n_steps && r < cnt) {
                    const int8_t* p;
                    if (D.n_slots == 0) {
                        p = reinterpret_cast<const int8_t*>(D.d_src) + idx;
                    } else {
                        // cif_deinterleaver.cpp:57-68: bit i comes from the CIF that is 15 - T[i mod 16] old, T = bitrev4
                        const int age = 15 - (int)(__brev((unsigned)idx & 15u) >> 28);
                        int slot = (int)D.newest_slot - age;
                        if (slot < 0) slot += (int)D.n_slots;
                        const int fr = slot / (int)D.cifs_per_frame, ci = slot - fr * (int)D.cifs_per_frame;
                        p = reinterpret_cast<const int8_t*>(D.d_src) + (size_t)fr * D.frame_stride + (size_t)ci * D.cif_stride + idx;
                    }
                    y = *p;
                    y = max(y, -127);          // soft-bit domain is [-127, +127] (viterbi_config.h:12-14)
                }
                // pack the 4 symbols of a step into one dword in every lane of the quad
                const int y0 = __builtin_amdgcn_mov_dpp(y, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
                const int y1 = __builtin_amdgcn_mov_dpp(y, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
                const int y2 = __builtin_amdgcn_mov_dpp(y, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
                const int y3 = __builtin_amdgcn_mov_dpp(y, 0xFF, 0xF, 0xF, true);   // [3,3,3,3]
                ypk = (y0 & 0xFF) | ((y1 & 0xFF) << 8) | ((y2 & 0xFF) << 16) | (y3 << 24);
            }

            // ---- 16 trellis steps ----
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int t = t0 + j;
                if (t < n_steps) {                                                   // wave-uniform
                    const int ysym = __builtin_amdgcn_readlane(ypk, 4 * j);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_lo, (int)metric);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_hi, (int)metric);
                    const int dot = __builtin_amdgcn_sdot4(sig, ysym, 0, false);
                    const uint32_t c0 = (lo + (uint32_t)(508 - dot)) & 0xFFFFu;      // u16 wrap like the reference core
                    const uint32_t c1 = (hi + (uint32_t)(508 + dot)) & 0xFFFFu;
                    const bool d = tie_rule ? (c1 <= c0) : (c0 > c1);
                    metric = d ? c1 : c0;
                    const uint64_t dec = __ballot(d);
                    const bool mine = (lane == (t & 63));                           // lane t mod 64 parks step t's word
                    vlo = mine ? (uint32_t)dec : vlo;
                    vhi = mine ? (uint32_t)(dec >> 32) : vhi;
                    const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)metric);
                    if (m0 >= V_RENORM) {                                            // rare, wave-uniform
                        uint32_t mn = metric;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
                        metric -= mn;
                        renorm_total += mn;
                    }
                }
            }
            // park 64 decision words (512 B, one coalesced store) every 64 steps and at the end
            if (((t0 + 16) & 63) == 0 || t0 + 16 >= n_steps) {
                const int base = (t0 + 15) & ~63;
                if (base + lane < n_steps) my_dec[base + lane] = ((uint64_t)vhi << 32) | vlo;
            }
        }
        const int end_state = (int)(D.end_state & 63u);
        const uint32_t end_metric = (uint32_t)__shfl((int)metric, end_state);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);       // decision words are re-read by this same wave: drain its stores first

        // ---- chain-back (Karn layout): 8-bit window, state in bits 7..2 ----
        const int n_bits = n_steps - 6;
        unsigned reg = (unsigned)end_state << 2;
        int cur_chunk = -1;
        uint32_t clo = 0, chi = 0;
        for (int bit = n_bits - 1; bit >= 0; bit--) {
            const int di = bit + 6;
            if ((di >> 6) != cur_chunk) {
                cur_chunk = di >> 6;
                const int w = cur_chunk * 64 + lane;
                uint64_t word = 0;
                if (w < n_steps) word = __builtin_nontemporal_load(&my_dec[w]);
                clo = (uint32_t)word; chi = (uint32_t)(word >> 32);
            }
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)clo, di & 63);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)chi, di & 63);
            const unsigned st = reg >> 2;
            const unsigned in = (unsigned)((((uint64_t)hi << 32) | lo) >> st) & 1u;
            reg = (reg >> 1) | (in << 7);
            if ((bit & 7) == 0 && lane == 0) {
                const int k = bit >> 3;
                obytes[k] = (unsigned char)(reg ^ ((D.flags & DABGPU_CW_RAW) ? 0u : prbs[k % PRBS_PERIOD]));   // descramble in the same pass
            }
        }
        __syncthreads();

        // ---- write-out + optional FIB CRC16 ----
        const int n_out = n_bits >> 3;
        unsigned char* out = reinterpret_cast<unsigned char*>(D.d_out);
        for (int k = lane; k < n_out; k += 64) out[k] = obytes[k];
        uint32_t crc_mask = 0;
        if (D.n_crc_blocks) {
            const int blk_bytes = n_out / (int)D.n_crc_blocks;
            bool ok = false;
            if (lane < (int)D.n_crc_blocks) {
                const unsigned char* fib = obytes + lane * blk_bytes;
                unsigned crc = 0xFFFFu;                                            // fic_decoder.cpp:19-31
                for (int i = 0; i < blk_bytes - 2; i++) {
                    crc ^= (unsigned)fib[i] << 8;
                    for (int q = 0; q < 8; q++) crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
                }
                crc ^= 0xFFFFu;
                const unsigned rx = ((unsigned)fib[blk_bytes - 2] << 8) | fib[blk_bytes - 1];
                ok = (rx == crc);
            }
            crc_mask = (uint32_t)__ballot(ok);
        }
        if (lane == 0) {
            dabgpu_cw_result R;
            R.path_error = renorm_total + end_metric;
            R.crc_ok_mask = crc_mask;
            R.n_out_bytes = (uint32_t)n_out;
            results[cw] = R;
        }
        __syncthreads();
    }
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_viterbi(const dabgpu_cw_desc* d_descs, int n_cw, uint64_t* d_scratch,
                                            size_t scratch_words_per_wave, int n_waves, int max_out_bytes,
                                            dabgpu_cw_result* d_results, int tie_rule, hipStream_t stream)
{
    using namespace dabgpu;
    const size_t lds = 25 * 8 * 2 + 512 + (size_t)((max_out_bytes + 15) & ~15);
    hipLaunchKernelGGL(viterbi_kernel, dim3((unsigned)n_waves), dim3(64), lds, stream,
                       d_descs, n_cw, d_scratch, scratch_words_per_wave, d_results, tie_rule);
    return hipGetLastError();
}

// ---- device-side descriptor builders (no host loops, no H2D per batch) ----
namespace dabgpu {

// FIC: 4 FIB groups per frame, PI_16 x 21 blocks, PI_15 x 3 blocks, tail (fic_decoder.cpp:53-84)
__global__ void fic_build_descs_kernel(dabgpu_cw_desc* descs, const int8_t* bits, size_t n_frames, size_t frame_stride,
                                       uint8_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * 4) return;
    const size_t fr = i >> 2, g = i & 3;
    dabgpu_cw_desc D = {};
    D.d_src = (uint64_t)(uintptr_t)(bits + fr * frame_stride + g * 2304);
    D.d_out = (uint64_t)(uintptr_t)(out + i * 96);
    D.n_steps = 768 + 6;
    D.seg_pi[0] = 16; D.seg_steps[0] = 32 * 21;
    D.seg_pi[1] = 15; D.seg_steps[1] = 32 * 3;
    D.n_crc_blocks = 3;
    descs[i] = D;
}

// MSC: codeword (ensemble e, cif c, sub-channel s) reads its logical frame through the time de-interleaver
// straight out of the history of demodulated frames (msc_decoder.cpp:46-75 + cif_deinterleaver.cpp:36-71)
__global__ void msc_build_descs_kernel(dabgpu_cw_desc* descs, const int8_t* hist, size_t n_ens, size_t ens_stride,
                                       int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* plans, int n_sub,
                                       uint8_t* out, size_t out_ens_stride, int cif_out_bytes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_ens = (size_t)4 * n_sub;
    if (i >= n_ens * per_ens) return;
    const size_t e = i / per_ens;
    const int rem = (int)(i - e * per_ens), c = rem / n_sub, s = rem - c * n_sub;
    const dabgpu_msc_plan P = plans[s];
    dabgpu_cw_desc D = {};
    D.d_src = (uint64_t)(uintptr_t)(hist + e * ens_stride + 9216 + (size_t)P.start_address * 64);
    D.d_out = (uint64_t)(uintptr_t)(out + e * out_ens_stride + (size_t)c * cif_out_bytes + P.out_offset);
    D.n_steps = P.n_steps;
    for (int k = 0; k < 4; k++) { D.seg_pi[k] = P.seg_pi[k]; D.seg_steps[k] = P.seg_steps[k]; }
    D.n_slots = 4 * hist_frames;
    D.newest_slot = 4 * newest_frame_slot + c;
    D.cifs_per_frame = 4;
    D.frame_stride = 230400;
    D.cif_stride = 55296;
    descs[i] = D;
}

// CIF_Deinterleaver::Deinterleave as a stand-alone gather (cif_deinterleaver.cpp:36-71); the decoder itself never
// materialises this buffer, this exists for callers of the CIF_Deinterleaver class
__global__ void cif_deinterleave_kernel(const int8_t* __restrict__ ring, int n_bits, int n_slots, int newest_slot,
                                        int8_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bits) return;
    const int age = 15 - (int)(__brev((unsigned)i & 15u) >> 28);
    int slot = newest_slot - age;
    if (slot < 0) slot += n_slots;
    out[i] = ring[(size_t)slot * n_bits + i];
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_cif_deinterleave(const int8_t* d_ring, int n_bits, int n_slots, int newest_slot,
                                                     int8_t* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(dabgpu::cif_deinterleave_kernel, dim3((unsigned)((n_bits + 255) / 256)), dim3(256), 0, stream,
                       d_ring, n_bits, n_slots, newest_slot, d_out);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_fic_build(dabgpu_cw_desc* d_descs, const int8_t* d_bits, size_t n_frames,
                                              size_t frame_stride, uint8_t* d_out, hipStream_t stream)
{
    const size_t n = n_frames * 4;
    hipLaunchKernelGGL(dabgpu::fic_build_descs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_descs, d_bits, n_frames, frame_stride, d_out);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_msc_build(dabgpu_cw_desc* d_descs, const int8_t* d_hist, size_t n_ens, size_t ens_stride,
                                              int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* d_plans, int n_sub,
                                              uint8_t* d_out, size_t out_ens_stride, int cif_out_bytes, hipStream_t stream)
{
    const size_t n = n_ens * 4 * (size_t)n_sub;
    hipLaunchKernelGGL(dabgpu::msc_build_descs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_descs, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, d_plans, n_sub, d_out,
                       out_ens_stride, cif_out_bytes);
    return hipGetLastError();
}
