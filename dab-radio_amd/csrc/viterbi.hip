// viterbi.hip -- gfx950 kernel for the channel-decode half of the hot path.
//
// One wavefront decodes one punctured K=7 rate-1/4 codeword (FIB group or sub-channel logical frame):
//   lanes             = the 64 trellis states, in a ROTATING layout: at step t (phase p = t mod 6) lane L holds state
//                       rotl6(L, p).  The two states of a butterfly (s, s+32) then sit in lanes L and L ^ (32 >> p), so the
//                       add-compare-select needs ONE cross-lane exchange per step and no LDS at all:
//                       v_permlane32_swap / v_permlane16_swap (gfx950) for p = 0,1, DPP row_ror:8 / bank-masked row_shl:4
//                       + row_shr:4 / quad_perm for p = 2..5.  Each lane keeps its own predecessor at branch cost e and
//                       takes the partner's at 1016 - e; the survivor becomes state rotl6(s,1) = the layout of phase p+1.
//   branch metric     = v_dot4_i32_i8 of the lane's sign pattern for this phase with 4 packed soft symbols,
//                       accumulated straight onto metric + 508
//   metrics           = u16 semantics (wrap) with the reference's renormalisation rule (dab_viterbi_decoder.cpp:31-41)
//   decisions         = kept TRANSPOSED: every lane shifts its own decision bit into a history register with one
//                       v_alignbit per step (sign bit of the candidate difference), 128 B (64 x u16) parked per 16 steps
//   de-puncturing     = index arithmetic on the kept-count tables (dab_viterbi_decoder.cpp:131-181), never materialised
//   time de-interleave= index arithmetic on a ring of past CIFs (cif_deinterleaver.cpp:36-71), never materialised
//   chain-back        = done in lane-index space (the survivor's lane changes one bit per step), all 64 lanes
//                       redundantly on the VALU -- the scalar unit is shared by the 32 waves of a CU and was the
//                       bottleneck of a scalar chain-back; one ds_bpermute per bit fetches the survivor lane's history
//                       word; MSB-first bytes, XOR with the energy-dispersal PRBS (additive_scrambler.h:16-35),
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
constexpr int VBLOCK = 48;                 // steps per block: multiple of 6 (layout phases), 8 (bytes), 16 (history words)

__device__ __forceinline__ int parity7(unsigned v) { return __builtin_popcount(v & 0x7Fu) & 1; }
__device__ __forceinline__ unsigned rotl6(unsigned v, int p) { return ((v << p) | (v >> (6 - p))) & 63u; }   // 0 <= p < 6
__device__ __forceinline__ unsigned rotr6(unsigned v, int p) { return ((v >> p) | (v << (6 - p))) & 63u; }

// kept-count of puncture vector PI_pi, 4-bit group g (0..7): the n-th extra kept bit of a 32-bit period goes to group
// bitrev3(n mod 8) (ETSI EN 300 401 table 13; same data as src/dab/constants/puncture_codes.h:42-67).
// PI_X (tail) == PI_8 restricted to 6 groups.
__device__ __forceinline__ int pi_count(int pi, int g) {
    const int ord = ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);
    return 1 + ((pi > ord) ? (((pi - 1 - ord) >> 3) + 1) : 0);
}

// value of `m` held by lane (lane ^ (32 >> P))
template <int P>
__device__ __forceinline__ uint32_t xchg(uint32_t m, int lane) {
    if constexpr (P == 0) {
        const auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false);
        return (lane & 32) ? r[0] : r[1];
    } else if constexpr (P == 1) {
        const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (P == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0x128, 0xF, 0xF, true);                 // row_ror:8
    } else if constexpr (P == 3) {
        const int a = __builtin_amdgcn_update_dpp((int)m, (int)m, 0x104, 0xF, 0x5, false);        // row_shl:4 -> banks 0,2
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)m, 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1,3
    } else if constexpr (P == 4) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0x4E, 0xF, 0xF, true);                  // quad_perm [2,3,0,1]
    } else {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xF, 0xF, true);                  // quad_perm [1,0,3,2]
    }
}

struct LaneConst {
    int sig[6];       // +sigma pattern of the lane's butterfly at phase p (4 x int8)
    int um[6];        // -1 when the lane holds the upper predecessor (state bit 5 = lane bit 5-p) at phase p, else 0
    int up1[6];       // -um (0 or 1), minus 1 under tie rule 1
};

// one trellis step at phase P: survivors, and the lane's decision bit (1 = upper predecessor s+32) shifted into `hist`.
// Nine instructions in the DPP phases: symbol broadcast, one dot, add, subtract, add with the lane exchange, 16-bit minimum, three for the decision
// bit.  What a lone wavefront (one receiver's frame: 76 code words on 1024 SIMDs) spends per step barely depends on their number or on the length
// of the chain between two metrics (~50 ns per step with 6 to 11 instructions and chains of 2 to 5: tools/abl_wave.py, profiles/r05/ab_notes.md).
template <int P, int TIE>
__device__ __forceinline__ void acs_step(uint32_t& metric, uint32_t& hist, const LaneConst& K, int ysym, int lane) {
    // own predecessor at branch cost e = 508 - dot, partner at 1016 - e = 508 + dot; the dot does not depend on the metrics
    const uint32_t ds = (uint32_t)__builtin_amdgcn_sdot4(K.sig[P], ysym, 0, false);
    const uint32_t m508 = metric + 508u;
    const uint32_t c_self = m508 - ds;
    const uint32_t c_part = xchg<P>(m508, lane) + ds;                    // (the DPP phases: one v_add_u32_dpp)
    // core model 0 (scalar core): the uint16_t sums wrap; model 1 (SIMD cores): adds_epu16, they saturate at 65535 (two v_min_u32)
    const uint16_t cs = (uint16_t)(TIE ? min(c_self, 65535u) : c_self), cp = (uint16_t)(TIE ? min(c_part, 65535u) : c_part);
    // lower lanes: upper predecessor = partner, chosen iff cp < cs (TIE 0) / <= (TIE 1);
    // upper lanes: upper predecessor = self,    chosen iff cs < cp (TIE 0) / <= (TIE 1)      (K.up1 holds the tie rule's -1)
    int d = (int)cp - (int)cs;
    d = (d ^ K.um[P]) + K.up1[P];
    hist = __builtin_amdgcn_alignbit(hist, (uint32_t)d, 31);          // hist = hist << 1 | sign(d)
    metric = (uint32_t)__builtin_elementwise_min(cs, cp);             // v_min_u16: one instruction for compare + select
}

// the reference's renormalisation (dab_viterbi_decoder.cpp:31-41): when metric[0] reaches the threshold, subtract the minimum
__device__ __forceinline__ void renormalise(uint32_t& metric, uint64_t& renorm_total) {
    uint32_t mn = metric;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    metric -= mn;
    renorm_total += mn;
}

template <int J, int TIE>
__device__ __forceinline__ void acs_at(uint32_t& metric, uint32_t& hist, const LaneConst& K, const int (&ypk)[3], int lane) {
    const int ysym = __builtin_amdgcn_readlane(ypk[J >> 4], 4 * (J & 15));
    acs_step<J % 6, TIE>(metric, hist, K, ysym, lane);
}

// 8 trellis steps J0 .. J0+7 of a block with no per-step scalar work: the caller has established that all 8 exist and that
// metric[0] (which grows by at most the largest branch cost, 1016, per step) cannot reach the renormalisation threshold
template <int J0, int TIE>
__device__ __forceinline__ void forward_group_fast(uint32_t& metric, uint32_t& hist, const LaneConst& K, const int (&ypk)[3], int t0,
                                                   int lane, uint16_t* my_dec16) {
    acs_at<J0, TIE>(metric, hist, K, ypk, lane);     acs_at<J0 + 1, TIE>(metric, hist, K, ypk, lane);
    acs_at<J0 + 2, TIE>(metric, hist, K, ypk, lane); acs_at<J0 + 3, TIE>(metric, hist, K, ypk, lane);
    acs_at<J0 + 4, TIE>(metric, hist, K, ypk, lane); acs_at<J0 + 5, TIE>(metric, hist, K, ypk, lane);
    acs_at<J0 + 6, TIE>(metric, hist, K, ypk, lane); acs_at<J0 + 7, TIE>(metric, hist, K, ypk, lane);
    if (((J0 + 7) & 15) == 15) my_dec16[(size_t)((t0 + J0 + 7) >> 4) * 64 + lane] = (uint16_t)hist;   // 16 steps x 64 lanes = 128 B
}

// the same 8 steps the careful way: bounds test and renormalisation check after every step.  Rare; a compact loop that is
// generic in the step position (phase by switch) keeps the instruction-cache footprint of the unrolled fast path small
template <int TIE>
__device__ __forceinline__ void forward_group_slow(uint32_t& metric, uint32_t& hist, uint64_t& renorm_total, const LaneConst& K,
                                                   int y0, int y1, int y2, int t0, int j0, int n_steps, int lane, uint16_t* my_dec16) {
#pragma nounroll
    for (int j = j0; j < j0 + 8 && t0 + j < n_steps; j++) {
        const int l4 = 4 * (j & 15);
        const int s0 = __builtin_amdgcn_readlane(y0, l4), s1 = __builtin_amdgcn_readlane(y1, l4), s2 = __builtin_amdgcn_readlane(y2, l4);
        const int ysym = (j < 16) ? s0 : ((j < 32) ? s1 : s2);
        switch (j % 6) {
        case 0: acs_step<0, TIE>(metric, hist, K, ysym, lane); break;
        case 1: acs_step<1, TIE>(metric, hist, K, ysym, lane); break;
        case 2: acs_step<2, TIE>(metric, hist, K, ysym, lane); break;
        case 3: acs_step<3, TIE>(metric, hist, K, ysym, lane); break;
        case 4: acs_step<4, TIE>(metric, hist, K, ysym, lane); break;
        default: acs_step<5, TIE>(metric, hist, K, ysym, lane); break;
        }
        const uint32_t m1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)metric);       // state 0 is lane 0 in every phase
        if (m1 >= V_RENORM) renormalise(metric, renorm_total);
        if ((j & 15) == 15) my_dec16[(size_t)((t0 + j) >> 4) * 64 + lane] = (uint16_t)hist;
    }
}

// up to 48 trellis steps; ypk[q] lane 4j = packed symbols of step t0 + 16 q + j.  Groups of 8 steps: unrolled and free of
// scalar work when all 8 exist and metric[0] cannot reach the threshold inside the group, else the careful loop
template <int TIE>
__device__ __forceinline__ void forward_block(uint32_t& metric, uint32_t& hist, uint64_t& renorm_total, const LaneConst& K,
                                              const int (&ypk)[3], int t0, int n_steps, int lane, uint16_t* my_dec16) {
#define VIT_GROUP(G)                                                                                                              \
    if (t0 + 8 * (G) < n_steps) {                                                         /* wave-uniform */                      \
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)metric);                                              \
        if (t0 + 8 * (G) + 8 <= n_steps && m0 + 8u * 1016u < V_RENORM)                                                          \
            forward_group_fast<8 * (G), TIE>(metric, hist, K, ypk, t0, lane, my_dec16);                                          \
        else                                                                                                                     \
            forward_group_slow<TIE>(metric, hist, renorm_total, K, ypk[0], ypk[1], ypk[2], t0, 8 * (G), n_steps, lane, my_dec16); \
    }
    VIT_GROUP(0) VIT_GROUP(1) VIT_GROUP(2) VIT_GROUP(3) VIT_GROUP(4) VIT_GROUP(5)
#undef VIT_GROUP
}

// up to 48 chain-back steps t = tb+47 .. tb (those with 6 <= t < n_steps).
// l4 = 4 * (lane holding the survivor after step t); acc collects the bits of the byte in flight
__device__ __forceinline__ void chainback_block(int& l4, uint32_t& acc, uint32_t& outreg, int tb, int n_steps, int lane,
                                                const uint16_t* my_dec16, unsigned char* obytes, const unsigned char* prbs, bool raw) {
    const int kb = tb >> 3;                       // bits tb-6 .. tb-1 belong to byte kb - 1, finished by the next (lower) block
    const int rel = (lane - (kb - 1)) & 63;       // the lane that parks byte (kb - 1 + c) is the one with rel == c
    uint32_t W = 0;
#pragma unroll
    for (int u = VBLOCK - 1; u >= 0; u--) {
        const int t = tb + u;
        if (t < n_steps && t >= 6) {                                             // wave-uniform
            if ((u & 15) == 15 || t == n_steps - 1)                              // entering history word t >> 4
                W = __builtin_nontemporal_load(&my_dec16[(size_t)(t >> 4) * 64 + lane]);
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)W);
            const uint32_t d = __builtin_amdgcn_ubfe(w, 15 - (u & 15), 1);
            const int q = 5 - (u % 6);                                           // the survivor's lane changes bit q (see header)
            l4 = (int)(((uint32_t)l4 & ~(4u << q)) | (d << (q + 2)));
            acc |= d << (7 - ((u + 2) & 7));                                     // bit (t-6) & 7, MSB first
            if (((u + 2) & 7) == 0) {                                            // byte (t - 6) >> 3 complete
                const int c = ((u - 6) >> 3) + 1;                                // byte index - (kb - 1): 1..6
                outreg = (rel == c) ? acc : outreg;
                acc = 0;
                const int k = kb - 1 + c;
                if ((k & 63) == 0) {                                             // bytes k .. k+63 parked (k = 0 ends the codeword): to LDS
                    const int kk = k + lane;
                    const unsigned char pb = raw ? (unsigned char)0 : prbs[kk % PRBS_PERIOD];
                    obytes[kk] = (unsigned char)(outreg ^ pb);
                }
            }
        }
    }
}

// The same 48 steps when all of them exist (6 <= tb, tb + 48 <= n_steps) and their three history words are already in registers.
// The survivor is ONE path: its lane is wave-uniform, so the walk is scalar -- v_readlane_b32 of the history word at the survivor's lane,
// s_bfe / s_lshl / s_or on the lane index and on the byte in flight -- instead of a ds_bpermute round trip and vector bit work per step,
// and free of the per-step bounds tests of chainback_block.  The caller loads the NEXT block's words before it calls this one: the walk
// used to stop at every 16th step for a global load nothing depended on.
__device__ __forceinline__ void chainback_block_full(int& l4, uint32_t& acc, uint32_t& outreg, int tb, int lane, const uint32_t (&W)[3],
                                                     unsigned char* obytes, const unsigned char* prbs, bool raw) {
    const int kb = tb >> 3;
    const int rel = (lane - (kb - 1)) & 63;
    int cur = __builtin_amdgcn_readfirstlane(l4) >> 2;
    uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc);
#pragma unroll
    for (int u = VBLOCK - 1; u >= 0; u--) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)W[u >> 4], cur);
        const uint32_t d = (w >> (15 - (u & 15))) & 1u;
        const int q = 5 - (u % 6);
        cur = (cur & ~(1 << q)) | (int)(d << q);
        a |= d << (7 - ((u + 2) & 7));
        if (((u + 2) & 7) == 0) {
            const int c = ((u - 6) >> 3) + 1;
            outreg = (rel == c) ? a : outreg;
            a = 0;
            const int k = kb - 1 + c;
            if ((k & 63) == 0) {
                const int kk = k + lane;
                const unsigned char pb = raw ? (unsigned char)0 : prbs[kk % PRBS_PERIOD];
                obytes[kk] = (unsigned char)(outreg ^ pb);
            }
        }
    }
    l4 = cur << 2;
    acc = a;
}

template <int TIE>
__global__ __launch_bounds__(64)
void viterbi_kernel(const dabgpu_cw_desc* __restrict__ descs, int n_cw, uint64_t* __restrict__ dec_scratch,
                    size_t scratch_words_per_wave, dabgpu_cw_result* __restrict__ results,
                    const dabgpu_vit_tables* __restrict__ tables, int n_first, dabgpu_cw_result* __restrict__ results_rest)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vsm[];
    uint16_t* pi_tab = reinterpret_cast<uint16_t*>(vsm);            // [25][8]: count | prefix << 8
    unsigned char* prbs = vsm + 25 * 8 * 2;                         // [512]
    unsigned long long* age_off = reinterpret_cast<unsigned long long*>(prbs + 512);   // [16] ring offsets of the 16 CIF ages
    unsigned char* obytes = reinterpret_cast<unsigned char*>(age_off + 16);             // decoded bytes of the current codeword

    const int lane = threadIdx.x;
    // ---- tables (host-computed constants of the context) ----
    for (int e = lane; e < 25 * 8; e += 64) pi_tab[e] = tables->pi_tab[e];
    for (int e = lane; e < 512; e += 64) prbs[e] = tables->prbs[e];
    __syncthreads();

    // ---- per-lane, per-phase constants: butterfly s' = rotl6(lane,p) & 31, symbol r expects +127 when
    // parity((2s') & G_r) (ViterbiBranchTable), error e = sum_r |branch_r - y_r| = 508 - sum_r sigma_r y_r ----
    const unsigned G[4] = {109u, 79u, 83u, 109u};                   // dab_viterbi_decoder.cpp:25
    LaneConst K;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const unsigned sp = rotl6((unsigned)lane, p) & 31u;
        int v = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sg = parity7((2u * sp) & G[r]) ? 1 : -1;
            v |= (sg & 0xFF) << (8 * r);
        }
        K.sig[p] = v;
        K.um[p] = ((lane >> (5 - p)) & 1) ? -1 : 0;
        K.up1[p] = -K.um[p] - (TIE != 0 ? 1 : 0);
    }

    uint16_t* my_dec16 = reinterpret_cast<uint16_t*>(dec_scratch + (size_t)blockIdx.x * scratch_words_per_wave);

    for (int cw = blockIdx.x; cw < n_cw; cw += gridDim.x) {
        const dabgpu_cw_desc D = descs[cw];
        const int n_steps = (int)D.n_steps;
        if (D.flags & DABGPU_CW_LANE_MAPPED) continue;        // decoded by vit_lanes_kernel in this call (hybrid MSC batches)
        // DABGPU_CW_DEPUNCTURED: the source holds all 4 mother symbols of every step (the caller de-punctured: any puncturing vectors, any
        // lengths, dab_viterbi_decoder.cpp:131-181), the segment tables are not read and any n_steps >= 1 is a codeword
        const bool full_rate = (D.flags & DABGPU_CW_DEPUNCTURED) != 0;
        if (n_steps < (full_rate ? 1 : 7)) {          // skipped work item (ring decode of an ensemble without a new frame)
            if (lane == 0) { dabgpu_cw_result R; R.path_error = 0; R.crc_ok_mask = 0; R.n_out_bytes = 0; *(cw < n_first ? results + cw : results_rest + (cw - n_first)) = R; }
            continue;
        }
        // segment boundaries in trellis steps; the 6 tail steps use PI_8 == PI_X.  Scalars, not arrays: an array indexed by the lane's segment number
        // ends up in private memory, and a kernel with a private segment pays for scratch at every dispatch
        const int sp0 = (int)D.seg_pi[0], sp1 = (int)D.seg_pi[1], sp2 = (int)D.seg_pi[2], sp3 = (int)D.seg_pi[3];
        const int se0 = (int)D.seg_steps[0], se1 = se0 + (int)D.seg_steps[1], se2 = se1 + (int)D.seg_steps[2], se3 = se2 + (int)D.seg_steps[3], se4 = se3 + 6;
        const int si1 = ((int)D.seg_steps[0] >> 3) * (8 + sp0), si2 = si1 + ((int)D.seg_steps[1] >> 3) * (8 + sp1),
                  si3 = si2 + ((int)D.seg_steps[2] >> 3) * (8 + sp2), si4 = si3 + ((int)D.seg_steps[3] >> 3) * (8 + sp3);

        // time de-interleaver (cif_deinterleaver.cpp:57-68): input bit i comes from the CIF that is 15 - bitrev4(i mod 16) CIFs
        // old; the byte offset of that CIF inside the frame-history ring depends on i mod 16 only -> 16-entry table per codeword
        __syncthreads();
        if (lane < 16) {
            unsigned long long off = 0;
            if (D.n_slots != 0) {
                const int age = 15 - (int)(__brev((unsigned)lane) >> 28);
                int slot = (int)D.newest_slot - age;
                if (slot < 0) slot += (int)D.n_slots;
                const int fr = slot / (int)D.cifs_per_frame, ci = slot - fr * (int)D.cifs_per_frame;
                off = (unsigned long long)fr * D.frame_stride + (unsigned long long)ci * D.cif_stride;
                if (D.flags & DABGPU_CW_CLASSED) off += (unsigned long long)lane * (D.cif_stride >> 4);    // class order: + class segment
            }
            age_off[lane] = off;
        }
        __syncthreads();
        const int8_t* src_base = reinterpret_cast<const int8_t*>(D.d_src);
        const unsigned ish = (D.n_slots != 0 && (D.flags & DABGPU_CW_CLASSED)) ? 4u : 0u;                  // ... + i / 16 inside the segment

        // ---- de-puncturing as a running index (dab_viterbi_decoder.cpp:131-181): lane L owns mother symbols 64 q + L of every
        // 48-step block, i.e. step offset so = (64 q + L) / 4, code bit r = L % 4.  Inside a puncturing segment the kept / dropped
        // pattern of that position repeats every 8 steps, so from block to block the input index just advances by
        // 6 x (8 + PI); only a lane that crosses into the next segment recomputes its position from the tables.
        int fidx[3], finc[3], frem[3], fkeep = 0;
        const int rbit = lane & 3;
        auto locate = [&](int q, int step) {
            if (full_rate) {                              // symbol r of step `step` is byte 4 step + r; nothing is punctured, no segments
                fidx[q] = 4 * step + rbit; finc[q] = 4 * VBLOCK; frem[q] = 0x40000000; fkeep |= 1 << q;
                return;
            }
            // (sums of increments, not a chain of selects: the compiler turns such a chain back into a table in private memory)
            const int c0 = step >= se0 ? -1 : 0, c1 = step >= se1 ? -1 : 0, c2 = step >= se2 ? -1 : 0, c3 = step >= se3 ? -1 : 0;
            const int sstart = (c0 & se0) + (c1 & (se1 - se0)) + (c2 & (se2 - se1)) + (c3 & (se3 - se2));
            const int send = se0 + (c0 & (se1 - se0)) + (c1 & (se2 - se1)) + (c2 & (se3 - se2)) + (c3 & (se4 - se3));
            const int pi = sp0 + (c0 & (sp1 - sp0)) + (c1 & (sp2 - sp1)) + (c2 & (sp3 - sp2)) + (c3 & (8 - sp3));
            const int in0 = (c0 & si1) + (c1 & (si2 - si1)) + (c2 & (si3 - si2)) + (c3 & (si4 - si3));
            const int sis = step - sstart;
            const uint16_t e = pi_tab[pi * 8 + (sis & 7)];
            const int cnt = e & 0xFF, pre = e >> 8;
            fidx[q] = in0 + (sis >> 3) * (8 + pi) + pre + rbit;
            finc[q] = (VBLOCK / 8) * (8 + pi);
            frem[q] = send - step;                        // steps from this one to the end of its segment (> 0 while inside)
            fkeep = (fkeep & ~(1 << q)) | ((rbit < cnt) ? (1 << q) : 0);
        };
#pragma unroll
        for (int q = 0; q < 3; q++) locate(q, 16 * q + (lane >> 2));

        uint32_t metric = (lane == (int)(D.start_state & 63u)) ? 0u : V_NONSTART;   // phase 0: lane = state
        uint64_t renorm_total = 0;
        uint32_t hist = 0;

        // ---- fetch + time de-interleave, one block AHEAD of the trellis: the loads of block t0 + 48 are in flight while block t0 runs (a lone
        // wavefront has nothing else to hide their latency behind: 33 blocks x ~2 us of a 1542-step codeword were spent waiting) ----
        int ynext[3];
        auto fetch = [&](int t0f) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int step = t0f + 16 * q + (lane >> 2);
                int y = 0;
                if (step < n_steps && ((fkeep >> q) & 1)) y = src_base[age_off[fidx[q] & 15] + (unsigned long long)((unsigned)fidx[q] >> ish)];
                ynext[q] = y;
                // this lane's position in the next block
                frem[q] -= VBLOCK;
                fidx[q] += finc[q];
                if (frem[q] <= 0 && step + VBLOCK < n_steps) locate(q, step + VBLOCK);      // crossed into another segment (rare)
            }
        };
        fetch(0);
        for (int t0 = 0; t0 < n_steps; t0 += VBLOCK) {
            int ypk[3], ycur[3] = {ynext[0], ynext[1], ynext[2]};
            if (t0 + VBLOCK < n_steps) fetch(t0 + VBLOCK);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int y = max(ycur[q], -127);      // soft-bit domain is [-127, +127] (viterbi_config.h:12-14)
                // pack the 4 symbols of a step into one dword in every lane of the quad
                const int y0 = __builtin_amdgcn_mov_dpp(y, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
                const int y1 = __builtin_amdgcn_mov_dpp(y, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
                const int y2 = __builtin_amdgcn_mov_dpp(y, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
                const int y3 = __builtin_amdgcn_mov_dpp(y, 0xFF, 0xF, 0xF, true);   // [3,3,3,3]
                ypk[q] = (y0 & 0xFF) | ((y1 & 0xFF) << 8) | ((y2 & 0xFF) << 16) | (y3 << 24);
            }
            forward_block<TIE>(metric, hist, renorm_total, K, ypk, t0, n_steps, lane, my_dec16);
        }
        if (n_steps & 15) {                                           // last, partial history word: left-align it
            hist <<= (16 - (n_steps & 15));
            my_dec16[(size_t)(n_steps >> 4) * 64 + lane] = (uint16_t)hist;
        }
        // layout after the last step (n_steps - 1) is phase n_steps % 6
        const int end_state = (int)(D.end_state & 63u);
        int l4 = (int)(rotr6((unsigned)end_state, n_steps % 6) << 2);
        const uint32_t end_metric = (uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)metric);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);       // history words are re-read by this same wave: drain its stores first

        // ---- chain-back over steps n_steps-1 .. 6 in aligned blocks of 48 ----
        const bool raw = (D.flags & DABGPU_CW_RAW) != 0;
        uint32_t acc = 0, outreg = 0;
        uint32_t wnext[3] = {0, 0, 0};
        bool have_next = false;
        for (int tb = ((n_steps - 1) / VBLOCK) * VBLOCK; tb >= 0; tb -= VBLOCK) {
            if (tb >= VBLOCK && tb + VBLOCK <= n_steps) {                       // a whole block above the first one (wave-uniform)
                uint32_t wc[3];
#pragma unroll
                for (int r = 0; r < 3; r++)
                    wc[r] = have_next ? wnext[r] : (uint32_t)__builtin_nontemporal_load(&my_dec16[(size_t)((tb >> 4) + r) * 64 + lane]);
                have_next = tb - VBLOCK >= VBLOCK;                               // the block below is whole too: its words, now
                if (have_next) {
#pragma unroll
                    for (int r = 0; r < 3; r++) wnext[r] = (uint32_t)__builtin_nontemporal_load(&my_dec16[(size_t)(((tb - VBLOCK) >> 4) + r) * 64 + lane]);
                }
                chainback_block_full(l4, acc, outreg, tb, lane, wc, obytes, prbs, raw);
            } else {
                have_next = false;
                chainback_block(l4, acc, outreg, tb, n_steps, lane, my_dec16, obytes, prbs, raw);
            }
        }
        __syncthreads();

        // ---- write-out + optional FIB CRC16 ----
        const int n_out = n_steps > 6 ? (n_steps - 6) >> 3 : 0;
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
            *(cw < n_first ? results + cw : results_rest + (cw - n_first)) = R;       // code words n_first .. report into a second array (the FIB groups of a frame decoded with its sub-channels)
        }
        __syncthreads();
    }
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_viterbi(const dabgpu_cw_desc* d_descs, int n_cw, uint64_t* d_scratch,
                                            size_t scratch_words_per_wave, int n_waves, int max_out_bytes,
                                            dabgpu_cw_result* d_results, int tie_rule, const dabgpu_vit_tables* d_tables,
                                            hipStream_t stream, int n_first, dabgpu_cw_result* d_results_rest)
{
    using namespace dabgpu;
    if (n_first <= 0 || !d_results_rest) { n_first = n_cw; d_results_rest = d_results; }
    const size_t lds = 25 * 8 * 2 + 512 + 16 * 8 + (size_t)((max_out_bytes + 63 + 15) & ~15);
    if (tie_rule)
        hipLaunchKernelGGL(viterbi_kernel<1>, dim3((unsigned)n_waves), dim3(64), lds, stream,
                           d_descs, n_cw, d_scratch, scratch_words_per_wave, d_results, d_tables, n_first, d_results_rest);
    else
        hipLaunchKernelGGL(viterbi_kernel<0>, dim3((unsigned)n_waves), dim3(64), lds, stream,
                           d_descs, n_cw, d_scratch, scratch_words_per_wave, d_results, d_tables, n_first, d_results_rest);
    return hipGetLastError();
}

// ---- device-side descriptor builders (no host loops, no H2D per batch) ----
namespace dabgpu {

// FIC: 4 FIB groups per frame, PI_16 x 21 blocks, PI_15 x 3 blocks, tail (fic_decoder.cpp:53-84)
// slots != nullptr: frame fr is slot slots[fr] of ensemble fr's frame-history ring (frame_stride = bytes per ensemble),
// a negative slot = no new frame for that ensemble (descriptor with n_steps = 0: the decoder skips it)
__device__ __forceinline__ void fic_build_one(size_t i, dabgpu_cw_desc* descs, const int8_t* bits, size_t n_frames, size_t frame_stride,
                                              uint8_t* out, const int32_t* slots)
{
    if (i >= n_frames * 4) return;
    const size_t fr = i >> 2, g = i & 3;
    dabgpu_cw_desc D = {};
    const int8_t* base = bits + fr * frame_stride;
    if (slots != nullptr) {
        const int sl = slots[fr];
        if (sl < 0) { descs[i] = D; return; }
        base += (size_t)sl * 230400;
    }
    D.d_src = (uint64_t)(uintptr_t)(base + g * 2304);
    D.d_out = (uint64_t)(uintptr_t)(out + i * 96);
    D.n_steps = 768 + 6;
    D.seg_pi[0] = 16; D.seg_steps[0] = 32 * 21;
    D.seg_pi[1] = 15; D.seg_steps[1] = 32 * 3;
    D.n_crc_blocks = 3;
    descs[i] = D;
}

__global__ void fic_build_descs_kernel(dabgpu_cw_desc* descs, const int8_t* bits, size_t n_frames, size_t frame_stride,
                                       uint8_t* out, const int32_t* slots)
{
    fic_build_one((size_t)blockIdx.x * blockDim.x + threadIdx.x, descs, bits, n_frames, frame_stride, out, slots);
}

// MSC: codeword (ensemble e, cif c, sub-channel s) reads its logical frame through the time de-interleaver
// straight out of the history of demodulated frames (msc_decoder.cpp:46-75 + cif_deinterleaver.cpp:36-71)
__device__ __forceinline__ void msc_build_one(size_t i, dabgpu_cw_desc* descs, const int8_t* hist, size_t n_ens, size_t ens_stride,
                                              int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* plans, int n_sub,
                                              uint8_t* out, size_t out_ens_stride, int cif_out_bytes, const int32_t* slots, int classed)
{
    const size_t per_ens = (size_t)4 * n_sub;
    if (i >= n_ens * per_ens) return;
    const size_t e = i / per_ens;
    const int rem = (int)(i - e * per_ens), c = rem / n_sub, s = rem - c * n_sub;
    const dabgpu_msc_plan P = plans[s];
    dabgpu_cw_desc D = {};
    if (slots != nullptr) {                           // per-ensemble ring position; negative = nothing new for this ensemble
        newest_frame_slot = slots[e];
        if (newest_frame_slot < 0) { descs[i] = D; return; }
    }
    // natural order: the sub-channel's first soft bit; class order: its first byte of class 0 (64 soft bits per CU = 4 per class)
    D.d_src = (uint64_t)(uintptr_t)(hist + e * ens_stride + 9216 + (size_t)P.start_address * (classed ? 4 : 64));
    if (classed) D.flags |= DABGPU_CW_CLASSED;
    D.d_out = (uint64_t)(uintptr_t)(out + e * out_ens_stride + (size_t)c * cif_out_bytes + P.out_offset);
    D.n_steps = P.n_steps;
    for (int k = 0; k < 4; k++) { D.seg_pi[k] = P.seg_pi[k]; D.seg_steps[k] = P.seg_steps[k]; }
    D.n_slots = 4 * hist_frames;
    D.newest_slot = 4 * newest_frame_slot + c;
    D.cifs_per_frame = 4;
    D.frame_stride = 230400;
    D.cif_stride = 55296;
    if (P.lane_mapped) D.flags |= DABGPU_CW_LANE_MAPPED;
    descs[i] = D;
}

__global__ void msc_build_descs_kernel(dabgpu_cw_desc* descs, const int8_t* hist, size_t n_ens, size_t ens_stride,
                                       int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* plans, int n_sub,
                                       uint8_t* out, size_t out_ens_stride, int cif_out_bytes, const int32_t* slots, int classed)
{
    msc_build_one((size_t)blockIdx.x * blockDim.x + threadIdx.x, descs, hist, n_ens, ens_stride, hist_frames, newest_frame_slot, plans, n_sub, out,
                  out_ens_stride, cif_out_bytes, slots, classed);
}

// both in one launch: descriptors 0 .. 4 n_ens n_sub - 1 = the sub-channels' code words, then the 4 n_ens FIB groups of the newest frames (a frame's
// FIC decoded with its sub-channels: a launch less between two trellis launches of a single receiver)
__global__ void msc_fic_build_descs_kernel(dabgpu_cw_desc* descs, const int8_t* hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                           int newest_frame_slot, const dabgpu_msc_plan* plans, int n_sub, uint8_t* out, size_t out_ens_stride,
                                           int cif_out_bytes, const int32_t* slots, int classed, const int8_t* fic_bits, uint8_t* fib_out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n_msc = n_ens * 4 * (size_t)n_sub;
    if (i < n_msc) msc_build_one(i, descs, hist, n_ens, ens_stride, hist_frames, newest_frame_slot, plans, n_sub, out, out_ens_stride, cif_out_bytes, slots, classed);
    else fic_build_one(i - n_msc, descs + n_msc, fic_bits, n_ens, ens_stride, fib_out, slots);
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
                                              size_t frame_stride, uint8_t* d_out, const int32_t* d_slots, hipStream_t stream)
{
    const size_t n = n_frames * 4;
    hipLaunchKernelGGL(dabgpu::fic_build_descs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_descs, d_bits, n_frames, frame_stride, d_out, d_slots);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_msc_build(dabgpu_cw_desc* d_descs, const int8_t* d_hist, size_t n_ens, size_t ens_stride,
                                              int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* d_plans, int n_sub,
                                              uint8_t* d_out, size_t out_ens_stride, int cif_out_bytes, const int32_t* d_slots,
                                              int classed, hipStream_t stream)
{
    const size_t n = n_ens * 4 * (size_t)n_sub;
    hipLaunchKernelGGL(dabgpu::msc_build_descs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_descs, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, d_plans, n_sub, d_out,
                       out_ens_stride, cif_out_bytes, d_slots, classed);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_msc_fic_build(dabgpu_cw_desc* d_descs, const int8_t* d_hist, size_t n_ens, size_t ens_stride,
                                                  int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* d_plans, int n_sub,
                                                  uint8_t* d_out, size_t out_ens_stride, int cif_out_bytes, const int32_t* d_slots,
                                                  int classed, const int8_t* d_fic_bits, uint8_t* d_fib_out, hipStream_t stream)
{
    const size_t n = n_ens * 4 * ((size_t)n_sub + 1);
    hipLaunchKernelGGL(dabgpu::msc_fic_build_descs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_descs, d_hist, n_ens,
                       ens_stride, hist_frames, newest_frame_slot, d_plans, n_sub, d_out, out_ens_stride, cif_out_bytes, d_slots, classed,
                       d_fic_bits, d_fib_out);
    return hipGetLastError();
}
