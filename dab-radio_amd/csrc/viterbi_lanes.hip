// viterbi_lanes.hip -- gfx950 kernels of the batch form of the channel decoder: ONE LANE per codeword.
//
// viterbi.hip spreads the 64 trellis states of one codeword over the lanes of a wavefront (12 VALU instructions and one
// cross-lane exchange per trellis step and codeword).  When a batch holds thousands of codewords with the SAME puncturing
// schedule (one sub-channel of many ensembles, the FIB groups of many frames) the transposed mapping is ~3.5x cheaper:
//   lane              = one codeword; a wavefront = a GROUP of 64 codewords that share n_steps / PI segments
//   metrics           = all 64 states of the lane's codeword in 32 VGPRs, two u16 metrics per register, biased by 0x8000 so
//                       that signed packed ops (v_pk_min_i16, v_pk_sub_i16 clamp) order them like the reference's unsigned
//                       u16 compares; wrap-around of v_pk_add_u16 = the reference's u16 wrap (dab_viterbi_decoder.cpp:31-41)
//   layout            = ROTATING: in layout L_q register idx holds the two states that differ in state bit q (idx = state
//                       with bit q removed).  A trellis step shifts every state bit up by one, so step phase q (= layout L_q,
//                       q = 0..4) reads X = M[i] (states b0,b1 < 32) and Y = M[i+16] (b0+32, b1+32) and writes the two packed
//                       survivors straight into N[2i], N[2i+1] of layout L_{q+1}: no cross-lane traffic, no shuffles, the
//                       same register pattern in every phase; only the branch-cost pairing differs.  In L_5 a register holds
//                       (s, s+32), i.e. both predecessors of one butterfly: that phase costs two extra ops per register.
//   branch costs      = e = 508 - (+-(y0+y3) +- y1 +- y2) (polynomials 0 and 3 are equal): 8 values per step, packed as
//                       C[sigma] = (e(sigma), e(sigma x f_q)) in 4 registers (4-5 v_dot4 + 4 v_pk_mad_i16, viterbi_pk16.h); the
//                       complement pattern is C[7 - sigma]
//   decisions         = sign bits of the saturated candidate differences, gathered with v_perm_b32 + v_bfi_b32 into
//                       2 dwords per step and lane, streamed to HBM two steps at a time ([step / 2][lane][4], 1 KB per store)
//   chain-back        = per lane, reading the lane's own decision words back (coalesced across the wavefront),
//                       MSB-first bytes, energy-dispersal XOR, FIB CRC16 -- same outputs as viterbi.hip
//   input             = vit_prep_*_kernel transpose the KEPT soft bits of a group (through the time de-interleaver) into
//                       [row][lane] dwords, 4 consecutive input bytes of the lane's codeword per row; the puncturing
//                       schedule is wave-uniform, so the forward pass de-punctures with scalar bookkeeping: the SALU tracks
//                       the input position of every step and hands the VALU one v_perm_b32 selector that picks the step's
//                       <= 4 kept bytes out of a two-row window and zeroes the punctured ones (dab_viterbi_decoder.cpp:131-181)
// Arithmetic and tie rules are those of viterbi.hip / oracle/dab_oracle_decode.c (bit-exact, incl. path_error).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabgpu_internal.h"
#include "viterbi_pk16.h"

namespace dabgpu {

// ---- decision gather: sign bytes of 32 difference registers -> 2 dwords ----
// The decision of new state n (layout L_QN after the step: register idx_QN(n), half = bit QN of n) goes to bit
// pos = rotr6(n, QN - 3) of the step's 64-bit word.  The half bit then is bit 3 of pos, so the 4 bytes of one v_perm_b32 (pos bits
// 4:3) come from two registers, and walking back one step (n' = n >> 1 | d << 5, QN' = QN - 1) changes exactly ONE bit of pos:
// bit (3 - QN) mod 6 := d.  The chain-back never computes a layout.
__host__ __device__ constexpr int vl_rotl6(int v, int s) { return ((v << s) | (v >> (6 - s))) & 63; }
__host__ __device__ constexpr int vl_idx(int n, int q) { return ((n >> (q + 1)) << q) | (n & ((1 << q) - 1)); }

template <int QN>
__device__ __forceinline__ void vl_gather(const s2 (&D)[32], uint32_t& w0, uint32_t& w1) {
    constexpr int S = (QN + 3) % 6;                                  // (QN - 3) mod 6
    uint32_t G[16];
#pragma unroll
    for (int g = 0; g < 16; g++) {
        // bits pos = 32 w + 8 j + k, g = 8 w + k: bytes j = 0,1 are the two halves of one register, j = 2,3 of another
        const int base = ((g >> 3) << 5) | (g & 7);
        const int r_lo = vl_idx(vl_rotl6(base, S), QN), r_hi = vl_idx(vl_rotl6(base | 16, S), QN);
        // selectors 8..11 of v_perm_b32 replicate the SIGN of bytes 1 / 3 / 5 / 7: every byte of G[g] is 0x00 or 0xFF
        G[g] = __builtin_amdgcn_perm(as_u32(D[r_hi]), as_u32(D[r_lo]), 0x0B0A0908u);
    }
    uint32_t a0 = G[0], a1 = G[8];
#pragma unroll
    for (int k = 1; k < 8; k++) {                                    // bit k of every byte <- G[k] (no shifts: the bytes are masks already)
        a0 = vl_bfi(0x01010101u << k, G[k], a0);
        a1 = vl_bfi(0x01010101u << k, G[8 + k], a1);
    }
    w0 = a0; w1 = a1;
}

// one trellis step in phase Q: layout L_Q (M) -> L_{(Q+1) mod 6} (N)
template <int Q, int TIE>
__device__ __forceinline__ void vl_step(const s2 (&M)[32], s2 (&N)[32], uint32_t ysym, int k508, uint32_t& w0, uint32_t& w1) {
    s2 C[8], D[32];
    {
        using map = vl_costmap<vl_flip(Q)>;                     // branch costs: viterbi_pk16.h
        constexpr int WF[4] = {map::wF(0, 1, 1, 1), map::wF(1, 1, 1, 1), map::wF(2, 1, 1, 1), map::wF(3, 1, 1, 1)};
        constexpr int WX[4] = {map::wX(0, 1, 1, 1), map::wX(1, 1, 1, 1), map::wX(2, 1, 1, 1), map::wX(3, 1, 1, 1)};
        vl_cost_table<vl_flip(Q), true>(ysym, WF, WX, k508, C);
    }
    if constexpr (Q < 5) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int s = vl_sigma(vl_ins_zero(i, Q));
            const s2 c1 = C[s], c2 = C[7 - s];
            const s2 t1 = addm<TIE>(M[i], c1), t2 = addm<TIE>(M[i + 16], c2);         // new state 2b:   lower + e | upper + (1016 - e)
            const s2 t3 = addm<TIE>(M[i], c2), t4 = addm<TIE>(M[i + 16], c1);         // new state 2b+1: lower + (1016 - e) | upper + e
            N[2 * i] = min16(t1, t2);
            N[2 * i + 1] = min16(t3, t4);
            // decision 1 = upper predecessor: TIE 0 iff lower > upper (sign of upper - lower); TIE 1 iff !(lower < upper)
            D[2 * i] = TIE ? satsub16(t1, t2) : satsub16(t2, t1);
            D[2 * i + 1] = TIE ? satsub16(t3, t4) : satsub16(t4, t3);
        }
    } else {
        // L_5: M[b] = (old[b], old[b+32]) and C[s] = (e, 1016 - e): broadcasting a half (op_sel, free) gives the packed
        // candidates of the new states (2b, 2b+1) directly
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const int s = vl_sigma(b);
            const s2 lower = addm<TIE>(__builtin_shufflevector(M[b], M[b], 0, 0), C[s]);          // (old[b] + e, old[b] + 1016 - e)
            const s2 upper = addm<TIE>(__builtin_shufflevector(M[b], M[b], 1, 1), C[7 - s]);      // (old[b+32] + 1016 - e, old[b+32] + e)
            N[b] = min16(lower, upper);
            D[b] = TIE ? satsub16(lower, upper) : satsub16(upper, lower);
        }
    }
    vl_gather<(Q + 1) % 6>(D, w0, w1);
    if constexpr (TIE != 0) { w0 = ~w0; w1 = ~w1; }
}

// the reference's renormalisation (dab_viterbi_decoder.cpp:31-41) for the lanes whose metric[0] reached the threshold
__device__ __forceinline__ void vl_renorm(s2 (&N)[32], uint32_t& total) {
    if ((int)N[0].x >= (int)(60455 - 32768)) {
        s2 red[16];
#pragma unroll
        for (int r = 0; r < 16; r++) red[r] = min16(N[r], N[r + 16]);
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1) {
#pragma unroll
            for (int r = 0; r < w; r++) red[r] = min16(red[r], red[r + w]);
        }
        s2 mn = red[0];
        mn = min16(mn, swap16(mn));
        const uint32_t mu = ((uint32_t)(uint16_t)mn.x) ^ 0x8000u;            // unbiased minimum
        const s2 sub = as_s2(mu | (mu << 16));
#pragma unroll
        for (int r = 0; r < 32; r++) N[r] = sub16(N[r], sub);
        total += mu;
    }
}

// ---- wave-uniform de-puncturing schedule (dab_viterbi_decoder.cpp:131-181) ----
// Inside a segment the kept / dropped pattern repeats every 8 steps (one 32-bit puncturing vector): step u of a period keeps
// cnt(u) of its 4 mother bits, always the first cnt (PI vectors only ever drop the tail of a 4-bit group), starting at input
// byte k = (bytes of the periods before) + prefix(u).  vit_sched_kernel writes, once per call and schedule, one (row, selector)
// pair per trellis step: row = k / 4 = the symbol row that holds the step's first kept byte, selector = the v_perm_b32 control
// that builds (y0, y1, y2, y3) from rows (row + 1 : row) -- byte r = input byte k + r for r < cnt, constant 0x00 (selector 0x0C)
// for the punctured ones.  The trellis kernels read the pairs with scalar loads, a block of steps ahead.
__device__ __forceinline__ uint2 vl_sched_entry(int t, const uint32_t* seg_pi, const uint32_t* seg_steps, uint32_t in_rows,
                                                const dabgpu_vit_tables* __restrict__ tables) {
    int sstart = 0, in0 = 0, pi = 8, k = 0;
    for (; k < 4; k++) {
        const int len = (int)seg_steps[k];
        if (t < sstart + len) { pi = (int)seg_pi[k]; break; }
        in0 += (len >> 3) * (8 + (int)seg_pi[k]);
        sstart += len;
    }
    // k == 4: the 6 tail steps, PI_X == PI_8 restricted to 6 groups (and the steps a prefetch reads past the end: their row is clamped)
    const int sis = t - sstart;
    const uint16_t e = tables->pi_tab[pi * 8 + (sis & 7)];
    const uint32_t cnt = e & 0xFFu, kk = (uint32_t)(in0 + (sis >> 3) * (8 + pi)) + (uint32_t)(e >> 8);
    const uint32_t keep = cnt >= 4u ? 0xFFFFFFFFu : ((1u << (8u * cnt)) - 1u);
    uint2 r;
    r.x = min(kk >> 2, in_rows - 2u);
    r.y = ((0x03020100u + (kk & 3u) * 0x01010101u) & keep) | (0x0C0C0C0Cu & ~keep);
    return r;
}

constexpr int VL_OBYTES = 100;              // codewords up to this many output bytes keep them in LDS for the CRC

// Workgroups of FOUR wavefronts = four groups: the hardware puts the wavefronts of a workgroup on the four SIMDs of one CU, so
// up to 1024 groups get a SIMD each.  Single-wavefront workgroups are placed one by one and a few per cent of them double up on
// a SIMD -- and one doubled SIMD (1.4 ms instead of 0.9) sets the time of the whole launch (tools/ubench/hwid_probe.hip).
// With at most one group per CU, or more groups than SIMDs, single-wavefront workgroups are ~5 % faster (no CU-level sharing).
// OCC = wavefronts per SIMD the register budget is set for: 4 (114 VGPRs, no scratch) or 5 (96 VGPRs, 22 spills): with 4097..5120
// groups five per SIMD finish in ONE round instead of a full round + a tail of lone wavefronts (4096 ensembles of the canonical multiplex
// = 4608 groups: 5.10 -> 4.92 ms)
template <int TIE, int VL_WAVES, int OCC = 4>
__global__ __launch_bounds__(64 * VL_WAVES) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void vit_lanes_kernel(const dabgpu_vit_group* __restrict__ groups, int n_groups, const dabgpu_cw_desc* __restrict__ descs,
                      const uint32_t* __restrict__ sym, uint32_t* __restrict__ dec, dabgpu_cw_result* __restrict__ results,
                      const dabgpu_vit_tables* __restrict__ tables, const uint2* __restrict__ sched)
{
    __shared__ unsigned char prbs[512];
    __shared__ unsigned short crc_tab[256];                        // CRC16 (x^16 + x^12 + x^5 + 1), one byte at a time
    // decoded bytes of CRC-checked codewords (FIB groups, 96 bytes): a lane re-reads its own row for the CRC -- from LDS, a byte load
    // from HBM per CRC step was a fifth of the FIC decode; rows of 25 dwords (odd) spread the lanes over the banks
    __shared__ unsigned char obuf[VL_WAVES][64][VL_OBYTES];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 512; e += 64 * VL_WAVES) prbs[e] = tables->prbs[e];
    for (int e = threadIdx.x; e < 256; e += 64 * VL_WAVES) {
        unsigned v = (unsigned)e << 8;
        for (int qq = 0; qq < 8; qq++) v = (v & 0x8000u) ? (((v << 1) ^ 0x1021u) & 0xFFFFu) : ((v << 1) & 0xFFFFu);
        crc_tab[e] = (unsigned short)v;
    }
    __syncthreads();
    const int wvi = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int group = (int)blockIdx.x * VL_WAVES + wvi;                                                           // wave-uniform
    if (group >= n_groups) return;

    const dabgpu_vit_group Gd = groups[group];
    const int T = (int)Gd.n_steps;
    const bool valid = lane < (int)Gd.count;
    const size_t cw = (size_t)Gd.first + (size_t)Gd.stride * (size_t)(valid ? lane : 0);
    const dabgpu_cw_desc Dd = descs[cw];
    const bool live = valid && Dd.n_steps != 0;              // n_steps == 0: skipped work item of a ring decode

    const uint32_t* grp_sym = sym + Gd.sym_off;              // [row][64]           (wave-uniform bases: scalar address + lane offset)
    uint32_t* grp_dec = dec + Gd.dec_off;                    // [step / 2][64][4]

    // ---- forward pass.  The LAST step runs in phase 5, so the metrics end in layout L_0.  n_steps = 8 m + 6 is even: the first,
    // partial block starts in phase q0 = 0, 2 or 4, and steps pair up (even phase, odd phase) into one 16-byte decision store ----
    const int q0 = (6 - T % 6) % 6;
    s2 M[32], N[32];
    {
        // start state ss in layout L_q0: register idx_q0(ss), half bit q0 of ss
        const uint32_t ss = Dd.start_state & 63u;
        const uint32_t r0 = ((ss >> (q0 + 1)) << q0) | (ss & ((1u << q0) - 1u)), h0 = (ss >> q0) & 1u;
        const uint32_t non = (5080u ^ 0x8000u), st = 0x8000u;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const uint32_t lo = ((uint32_t)r == r0 && h0 == 0) ? st : non, hi = ((uint32_t)r == r0 && h0 == 1) ? st : non;
            M[r] = as_s2(lo | (hi << 16));
        }
    }
    uint32_t total = 0;
    int k508 = 508;
    asm volatile("" : "+v"(k508));                          // one register for the accumulator constant of the cost dot products
    const uint2* sch = sched + Gd.sched_off;                 // (row, selector) per trellis step: scalar loads
    const uint32_t lane4 = 4u * (uint32_t)lane, lane16 = 16u * (uint32_t)lane;
    // two trellis steps t, t + 1 (t even) in phases Q, Q + 1: M -> N -> M
#define VL_PAIR(Q, TT, YA, YB)                                                                        \
    {                                                                                                 \
        uint4 wv;                                                                                     \
        vl_step<Q, TIE>(M, N, YA, k508, wv.x, wv.y);                                                        \
        vl_renorm(N, total);                                                                          \
        vl_step<(Q) + 1, TIE>(N, M, YB, k508, wv.z, wv.w);                                                  \
        vl_renorm(M, total);                                                                          \
        __builtin_nontemporal_store(u4v{wv.x, wv.y, wv.z, wv.w},                                      \
                                    reinterpret_cast<u4v*>(reinterpret_cast<char*>(grp_dec + (size_t)((TT) >> 1) * 256) + lane16)); \
    }
    // the packed symbols of step TT: two rows of the group's symbol area (scalar base + lane offset) and the selector
#define VL_ROWS(LO, HI, SEL, TT)                                                                      \
    {                                                                                                 \
        const uint2 e_ = sch[TT];                                                                     \
        const char* rp_ = reinterpret_cast<const char*>(grp_sym + (size_t)e_.x * 64);                 \
        LO = *reinterpret_cast<const uint32_t*>(rp_ + lane4);                                         \
        HI = *reinterpret_cast<const uint32_t*>(rp_ + 256 + lane4);                                   \
        SEL = e_.y;                                                                                   \
    }
    int t = 0;
    if (q0 == 2) {
        uint32_t a0, a1, b0, b1, sa, sb;
        VL_ROWS(a0, a1, sa, t) VL_ROWS(b0, b1, sb, t + 1)
        VL_PAIR(2, t, __builtin_amdgcn_perm(a1, a0, sa), __builtin_amdgcn_perm(b1, b0, sb))
        t += 2;
    }
    if (q0 != 0) {
        uint32_t a0, a1, b0, b1, sa, sb;
        VL_ROWS(a0, a1, sa, t) VL_ROWS(b0, b1, sb, t + 1)
        VL_PAIR(4, t, __builtin_amdgcn_perm(a1, a0, sa), __builtin_amdgcn_perm(b1, b0, sb))
        t += 2;
    }
    uint32_t nlo[6], nhi[6], nsel[6];
#pragma unroll
    for (int q = 0; q < 6; q++) VL_ROWS(nlo[q], nhi[q], nsel[q], t + q)
    for (; t < T; t += 6) {
        uint32_t y[6];
#pragma unroll
        for (int q = 0; q < 6; q++) y[q] = __builtin_amdgcn_perm(nhi[q], nlo[q], nsel[q]);
#pragma unroll
        for (int q = 0; q < 6; q++) VL_ROWS(nlo[q], nhi[q], nsel[q], t + 6 + q)               // one block ahead (the table runs a block past the end)
        VL_PAIR(0, t, y[0], y[1])
        VL_PAIR(2, t + 2, y[2], y[3])
        VL_PAIR(4, t + 4, y[4], y[5])
    }
#undef VL_ROWS
#undef VL_PAIR

    // ---- end metric (layout L_0: register es >> 1, half es & 1) ----
    const uint32_t es = Dd.end_state & 63u;
    uint32_t endm = 0;
#pragma unroll
    for (int r = 0; r < 32; r++) endm = ((es >> 1) == (uint32_t)r) ? as_u32(M[r]) : endm;
    endm = (((es & 1u) ? (endm >> 16) : endm) & 0xFFFFu) ^ 0x8000u;

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);               // decision words are re-read by this same lane: drain its stores first (same CU, same
                                                 // L2: no write-back is needed -- an agent-scope release writes the whole L2 back)

    // ---- chain-back over steps T-1 .. 6 (dab_viterbi_decoder.cpp:124-129): decoded bit t-6 = decision of the survivor at step t ----
    const bool raw = (Dd.flags & DABGPU_CW_RAW) != 0;
    unsigned char* out = reinterpret_cast<unsigned char*>(Dd.d_out);
    const int n_out = (T - 6) >> 3;
    const bool crc_lds = Dd.n_crc_blocks != 0 && n_out <= VL_OBYTES;
    unsigned char* const ob = &obuf[wvi][lane][0];
    // The walk is done on pos = rotr6(state, QN - 3) (vl_gather).  Chunks of 24 steps: n_steps = 8 m + 6 and the last step ends in
    // layout L_0, so every chunk starts in layout L_0 at bit 7 of a byte -- phases and byte boundaries are compile-time.
    // One chunk of decision words (24 steps = 12 sixteen-byte loads per lane) is fetched, then walked.  The earlier version fetched
    // the next chunk while walking this one; its second set of 48 registers spilled under the 128-VGPR budget of the forward pass,
    // and the walk of all resident wavefronts is bound by HBM throughput, not by the latency of one wavefront's loads.
    uint32_t pos = ((es >> 3) | (es << 3)) & 63u;                  // rotr6(es, 3): layout L_0 after the last step
    constexpr int CB = 24;
    uint32_t cx[CB], cy[CB];
    for (int th = T - 1; th >= 6; th -= CB) {
        // words of steps th - u, u = 0..23, th odd: pair (th - 2 v) >> 1 holds step th - 2 v - 1 in .x .y and step th - 2 v in .z .w
#pragma unroll
        for (int v = 0; v < CB / 2; v++) {
            const int pr = (th >> 1) - v;
            const u4v q4 = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(grp_dec + (size_t)(pr < 0 ? 0 : pr) * 256) + lane16));
            cx[2 * v] = q4.z; cy[2 * v] = q4.w; cx[2 * v + 1] = q4.x; cy[2 * v + 1] = q4.y;
        }
#pragma unroll
        for (int ub = 0; ub < CB; ub += 8) {
            if (th - ub >= 6) {                                            // wave-uniform; steps come in whole bytes
                uint32_t acc = 0;
#pragma unroll
                for (int u = ub; u < ub + 8; u++) {
                    const uint32_t word = (pos & 32u) ? cy[u] : cx[u];
                    const uint32_t d = (word >> (pos & 31u)) & 1u;
                    const int b = (3 + u) % 6;                             // (3 - QN) mod 6 with QN = (-u) mod 6
                    pos = (pos & ~(1u << b)) | (d << b);
                    acc |= d << (u - ub);                                  // bit index (t - 6) & 7 = 7 - (u - ub), MSB first
                }
                const int k = (th - ub - 7 - 6) >> 3;                      // byte of bits t-6 for t = th-ub-7 .. th-ub
                const unsigned char pb = raw ? (unsigned char)0 : prbs[k % VL_PRBS];
                if (live) out[k] = (unsigned char)(acc ^ pb);
                if (crc_lds) ob[k] = (unsigned char)(acc ^ pb);
            }
        }
    }

    // ---- optional FIB CRC16 (fic_decoder.cpp:19-31,103-116) over the lane's own bytes + result record ----
    uint32_t crc_mask = 0;
    if (live && Dd.n_crc_blocks) {
        if (!crc_lds) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
        }
        const int blk_bytes = n_out / (int)Dd.n_crc_blocks;
        for (int b = 0; b < (int)Dd.n_crc_blocks && b < 32; b++) {
            unsigned crc = 0xFFFFu, rx;
            if (crc_lds) {
                const unsigned char* fib = ob + b * blk_bytes;
                for (int i = 0; i < blk_bytes - 2; i++) crc = ((crc << 8) ^ crc_tab[((crc >> 8) ^ fib[i]) & 0xFFu]) & 0xFFFFu;
                rx = ((unsigned)fib[blk_bytes - 2] << 8) | fib[blk_bytes - 1];
            } else {
                const unsigned char* fib = out + b * blk_bytes;
                for (int i = 0; i < blk_bytes - 2; i++)
                    crc = ((crc << 8) ^ crc_tab[((crc >> 8) ^ __builtin_nontemporal_load(&fib[i])) & 0xFFu]) & 0xFFFFu;
                rx = ((unsigned)__builtin_nontemporal_load(&fib[blk_bytes - 2]) << 8) | __builtin_nontemporal_load(&fib[blk_bytes - 1]);
            }
            crc ^= 0xFFFFu;
            if (rx == crc) crc_mask |= 1u << b;
        }
    }
    if (valid) {
        dabgpu_cw_result R;
        R.path_error = live ? (uint64_t)total + endm : 0;
        R.crc_ok_mask = crc_mask;
        R.n_out_bytes = live ? (uint32_t)n_out : 0u;
        *reinterpret_cast<dabgpu_cw_result*>(reinterpret_cast<char*>(results + cw) + Gd.res_delta) = R;
    }
}

// ---- input transposition: the kept soft bits of a group's codewords -> [row][lane] dwords ----
// Row j of lane L = input bytes 4 j .. 4 j + 3 of L's codeword, read through the time de-interleaver
// (cif_deinterleaver.cpp:57-68) where the codeword has a CIF ring, -128 clamped to -127 (viterbi_config.h:12-14), zero past the
// end of the input.  None of these kernels knows the puncturing schedule any more: that is the trellis kernel's scalar bookkeeping.

// -128 -> -127 in each of the four bytes (no packed byte max on gfx950): flag the bytes equal to 0x80 exactly, add the flag
__device__ __forceinline__ uint32_t vl_clamp4(uint32_t w) {
    const uint32_t t = w ^ 0x80808080u;                                      // zero byte <=> -128
    const uint32_t nz = ((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t;               // bit 7 of a byte set <=> the byte of t is non-zero
    return w + ((~nz & 0x80808080u) >> 7);                                   // 0x80 + 1 = 0x81: no carry between bytes
}

// general form (any alignment, any ring geometry, natural or class order): byte loads through the 16 ring offsets.
// grid (n_groups, ceil(max in_rows / 64)), 256 threads: wave v handles codewords v, v + 4, ... of the group, lane = row of the tile
__global__ __launch_bounds__(256)
void vit_prep_kernel(const dabgpu_vit_group* __restrict__ groups, const dabgpu_cw_desc* __restrict__ descs, uint32_t* __restrict__ sym)
{
    __shared__ uint32_t tile[VL_TILE][65];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const dabgpu_vit_group Gd = groups[blockIdx.x];
    const int j0 = blockIdx.y * VL_TILE;
    if (j0 >= (int)Gd.in_rows) return;
    const unsigned n_in = dabgpu_vit_in_bytes(Gd.seg_pi, Gd.seg_steps);
    const unsigned i0 = 4u * (unsigned)(j0 + lane);
    for (int c = wv; c < 64; c += 4) {
        uint32_t packed = 0;
        const bool have = c < (int)Gd.count;                                   // wave-uniform
        const dabgpu_cw_desc Dd = descs[(size_t)Gd.first + (size_t)Gd.stride * (size_t)(have ? c : 0)];
        // time de-interleaver (cif_deinterleaver.cpp:57-68): input bit i lives in the CIF that is 15 - bitrev4(i mod 16) CIFs old;
        // lane L holds the ring offset of class i mod 16 == L mod 16 (the ring of one ensemble is < 4 GiB)
        uint32_t aoff = 0;
        unsigned ish = 0;                                                      // class order (DABGPU_CW_CLASSED): bit i sits at class base + i / 16
        if (Dd.n_slots != 0) {
            const int age = 15 - (int)(__brev((unsigned)lane & 15u) >> 28);
            int slot = (int)Dd.newest_slot - age;
            if (slot < 0) slot += (int)Dd.n_slots;
            const int fr = slot / (int)Dd.cifs_per_frame, ci = slot - fr * (int)Dd.cifs_per_frame;
            aoff = (uint32_t)fr * Dd.frame_stride + (uint32_t)ci * Dd.cif_stride;
            if (Dd.flags & DABGPU_CW_CLASSED) { aoff += ((unsigned)lane & 15u) * (Dd.cif_stride >> 4); ish = 4; }
        }
        const int8_t* src = reinterpret_cast<const int8_t*>(Dd.d_src);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned i = i0 + (unsigned)r;
            const uint32_t off = (uint32_t)__shfl((int)aoff, (int)(i & 15u));
            if (have && Dd.n_steps != 0 && i < n_in) {
                int yv = src[(size_t)off + (i >> ish)];
                yv = max(yv, -127);                                            // soft-bit domain is [-127, +127] (viterbi_config.h:12-14)
                packed |= ((uint32_t)yv & 0xFFu) << (8 * r);
            }
        }
        tile[lane][c] = packed;
    }
    __syncthreads();
    uint32_t* dst = sym + Gd.sym_off;
    for (int s = wv; s < VL_TILE; s += 4) {
        if (j0 + s < (int)Gd.in_rows) dst[(size_t)(j0 + s) * 64 + lane] = tile[s][lane];
    }
}

// direct form (FIC): every codeword of the group is a contiguous run of soft bits at a 16-byte aligned address.  A tile = 64 rows =
// 256 input bytes of each of the 64 codewords: 1024 sixteen-byte chunks, all in flight at once, land in LDS rows of 65 dwords
// (conflict-free both ways) and leave transposed, one coalesced 256-byte row store per wave and row.
// grid (n_groups, ceil(max in_rows / 64)), 256 threads
__global__ __launch_bounds__(256)
void vit_prep_direct_kernel(const dabgpu_vit_group* __restrict__ groups, const dabgpu_cw_desc* __restrict__ descs, uint32_t* __restrict__ sym)
{
    __shared__ uint32_t rows[64][65];                                          // [codeword][dword of the tile]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const dabgpu_vit_group Gd = groups[blockIdx.x];
    const int j0 = blockIdx.y * VL_TILE;
    if (j0 >= (int)Gd.in_rows) return;
    const int n_in = (int)dabgpu_vit_in_bytes(Gd.seg_pi, Gd.seg_steps);       // a multiple of 4
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    u4v v[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int q = tid + 256 * h, c = q >> 4, ch = q & 15;
        v[h] = u4v{0u, 0u, 0u, 0u};
        if (c < (int)Gd.count) {
            const dabgpu_cw_desc* Dd = descs + ((size_t)Gd.first + (size_t)Gd.stride * (size_t)c);
            const int o = 4 * j0 + 16 * ch;
            if (Dd->n_steps != 0 && o < n_in) {
                const gptr p = (gptr)(uintptr_t)Dd->d_src + o;
                if (o + 16 <= n_in) v[h] = *(const __attribute__((address_space(1))) u4v*)p;
                else {                                                     // the last chunk of the codeword: whole dwords only
                    const __attribute__((address_space(1))) uint32_t* pw = (const __attribute__((address_space(1))) uint32_t*)p;
                    v[h].x = pw[0];
                    if (o + 8 <= n_in) v[h].y = pw[1];
                    if (o + 12 <= n_in) v[h].z = pw[2];
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int q = tid + 256 * h, c = q >> 4, ch = q & 15;
        rows[c][4 * ch] = v[h].x; rows[c][4 * ch + 1] = v[h].y; rows[c][4 * ch + 2] = v[h].z; rows[c][4 * ch + 3] = v[h].w;
    }
    __syncthreads();
    uint32_t* dst = sym + Gd.sym_off;
    for (int s = wv; s < VL_TILE; s += 4) {                                    // lane = codeword
        if (j0 + s < (int)Gd.in_rows) dst[(size_t)(j0 + s) * 64 + lane] = vl_clamp4(rows[lane][s]);
    }
}

// MSC form, natural order: lane 4 k + c of a group = CIF c of ensemble k (16 ensembles), all read through a frame-history
// ring with 4 CIFs per frame.  Output CIFs c = 0..3 with ages 0..15 touch the 19 ring slots 4 nf - 15 .. 4 nf + 3; per ensemble
// the tile's 256-byte range of those 19 rows is staged in LDS with 16-byte loads (the byte-granular gather of vit_prep_kernel
// costs one address-coalescer slot per BYTE), then thread (row, c) picks its 4 soft bits from LDS.
// grid (n_groups, ceil(max in_rows / 64)), 256 threads = 64 rows x 4 CIFs
constexpr int VR_ROWS = 19, VR_CHUNKS = 16, VR_PITCH = VR_CHUNKS * 16 + 16;
__global__ __launch_bounds__(256)
void vit_prep_ring4_kernel(const dabgpu_vit_group* __restrict__ groups, const dabgpu_cw_desc* __restrict__ descs, uint32_t* __restrict__ sym)
{
    __shared__ uint32_t tile[VL_TILE][65];
    __shared__ __attribute__((aligned(16))) unsigned char rows[2][VR_ROWS][VR_PITCH];
    const int tid = threadIdx.x, rowi = tid & 63, c = tid >> 6;
    const dabgpu_vit_group Gd = groups[blockIdx.x];
    const int j0 = blockIdx.y * VL_TILE;
    if (j0 >= (int)Gd.in_rows) return;
    const int n_in = (int)dabgpu_vit_in_bytes(Gd.seg_pi, Gd.seg_steps);       // EEP: the sub-channel size; UEP: less, the rest is padding
    const int i_lo = 4 * j0, i_hi = min(i_lo + 4 * VL_TILE, n_in);            // the tile's byte range
    // LDS address of the thread's r-th byte: row = c - age + 15, column = i - i_lo  (the same for every ensemble)
    int lds_off[4];
    bool in[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const unsigned i = (unsigned)(i_lo + 4 * rowi + r);
        const int age = 15 - (int)(__brev(i & 15u) >> 28);
        lds_off[r] = (c - age + 15) * VR_PITCH + (4 * rowi + r);
        in[r] = (int)i < i_hi;
    }
    // staging loads: chunk q = tid, tid + 256 of the 19 x 16 sixteen-byte chunks of one ensemble
    auto fetch = [&](int k, uint4 (&v)[2]) {
        const bool have = 4 * k < (int)Gd.count;
        const dabgpu_cw_desc Dd = descs[(size_t)Gd.first + (size_t)Gd.stride * (size_t)(have ? 4 * k : 0)];
        const bool on = have && Dd.n_steps != 0;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(Dd.d_src);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = tid + 256 * h, row = q >> 4, ch = q & 15;
            v[h] = make_uint4(0, 0, 0, 0);
            if (on && row < VR_ROWS && i_lo + 16 * ch < i_hi) {
                int slot = (int)Dd.newest_slot - 15 + row;                 // lane 4 k is CIF 0: its newest slot is 4 nf
                if (slot < 0) slot += (int)Dd.n_slots;
                if (slot >= (int)Dd.n_slots) slot -= (int)Dd.n_slots;
                const size_t off = (size_t)(slot >> 2) * Dd.frame_stride + (size_t)(slot & 3) * Dd.cif_stride + (size_t)(i_lo + 16 * ch);
                v[h] = *reinterpret_cast<const uint4*>(src + off);
            }
        }
    };
    auto stash = [&](int buf, const uint4 (&v)[2]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int q = tid + 256 * h, row = q >> 4, ch = q & 15;
            if (row < VR_ROWS) *reinterpret_cast<uint4*>(&rows[buf][row][16 * ch]) = v[h];
        }
    };
    // four ensembles of loads in flight: with one, every ensemble costs a full memory round trip per workgroup
    uint4 v[4][2];
    fetch(0, v[0]); fetch(1, v[1]); fetch(2, v[2]); fetch(3, v[3]);
    stash(0, v[0]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k + 4 < 16) fetch(k + 4, v[k & 3]);
        const unsigned char* rb = &rows[k & 1][0][0];
        uint32_t packed = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (in[r]) {
                int yv = (int)(signed char)rb[lds_off[r]];
                yv = max(yv, -127);                                        // soft-bit domain is [-127, +127] (viterbi_config.h:12-14)
                packed |= ((uint32_t)yv & 0xFFu) << (8 * r);
            }
        }
        tile[rowi][4 * k + c] = packed;
        if (k + 1 < 16) stash((k + 1) & 1, v[(k + 1) & 3]);
        __syncthreads();
    }
    uint32_t* dst = sym + Gd.sym_off;
    const int lane = tid & 63, wv = tid >> 6;
    for (int s = wv; s < VL_TILE; s += 4) {
        if (j0 + s < (int)Gd.in_rows) dst[(size_t)(j0 + s) * 64 + lane] = tile[s][lane];
    }
}

// ---- MSC form, history in time-interleaver class order (DABGPU_CW_CLASSED) ----
// In class order the bits an output CIF takes from one history row are contiguous: class k of output CIF q lives in row
// q - age(k) + 15 at [k * cif_stride / 16 + i / 16].  A workgroup owns a QUARTER group (4 ensembles x 4 CIFs = 16 lanes) and walks
// its symbol rows in tiles of 256 = 1024 input bytes per lane = 64 bytes of each of the 4 x 4 x 16 class streams (ensemble x output
// CIF x class).  The streams are fetched ONE 64-BYTE MEMORY LINE AT A TIME into per-stream LDS rings of two lines: a tile loads only
// the line the previous tiles have not brought in yet -- every history byte crosses HBM -> LDS once -- and the next tile's line is in
// flight while this tile is transposed: a thread takes the same four columns of four classes (four LDS dwords), transposes the
// 4 x 4 bytes with eight v_perm_b32 and stores four row dwords; the 16 threads (ensemble, CIF) of a quarter group fill one 64-byte
// sector of a row.  Group and ring descriptors are read once per workgroup (scalar loads).
// grid (4 n_groups), 256 threads: thread = (stream tid / 4 of an ensemble, 16-byte chunk tid % 4 of a line) when loading
constexpr int VC_ROWS = 256, VC_ENS = 4;
constexpr int VC_PITCH = 2 * 64 + 16;                                      // bytes between the rings of two classes (36 dwords)
constexpr int VC_CIF_PITCH = 16 * VC_PITCH + 16;                           // bytes between two output CIFs: 580 dwords = 4 mod 64 banks
constexpr int VC_ENS_PITCH = 4 * VC_CIF_PITCH;                             // bytes between two ensembles: 2320 dwords = 16 mod 64, so the
                                                                           // dword reads of a half wave fall two per bank
__global__ __launch_bounds__(256)
void vit_prep_ring4c_kernel(const dabgpu_vit_group* __restrict__ groups, const dabgpu_cw_desc* __restrict__ descs, uint32_t* __restrict__ sym,
                            uint32_t groups_per_sub, uint32_t n_lane_sub)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[VC_ENS * VC_ENS_PITCH];       // [ensemble][output CIF][class][2 lines]
    const int tid = threadIdx.x;
    // Which (group, quarter) this workgroup gathers.  groups_per_sub != 0 (MSC: groups[li * groups_per_sub + gq] = sub-channel li of
    // ensembles 16 gq ..): XCD-aware order.  The sub-channels of one quarter group of ensembles read neighbouring pieces of the same
    // class rows -- the 64-byte lines at both ends of a piece are shared with the neighbour -- and workgroups are dealt round-robin to
    // the 8 XCDs, each with its own L2: unit (gq, quarter) u goes to XCD u mod 8 with its n_lane_sub sub-channels back to back.
    unsigned g_idx = blockIdx.x >> 2;
    int qg = (int)(blockIdx.x & 3);
    if (groups_per_sub != 0) {
        const unsigned x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const unsigned li = j % n_lane_sub, u = (j / n_lane_sub) * 8 + x;
        if (u >= 4 * groups_per_sub) return;
        g_idx = li * groups_per_sub + (u >> 2);
        qg = (int)(u & 3);
    }
    const dabgpu_vit_group Gd = groups[g_idx];
    const int n_in = (int)dabgpu_vit_in_bytes(Gd.seg_pi, Gd.seg_steps);       // input soft bits the decoder consumes (EEP: the sub-channel
                                                                               // size; UEP: less, the rest is padding)
    // the four ensembles of this quarter group (uniform descriptors: scalar loads, once per workgroup)
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    gptr src[VC_ENS];
    unsigned newest[VC_ENS], n_slots = 16, frame_stride = 0, cif_stride = 0;
    bool on[VC_ENS], any = false;
    int phi = 0;
#pragma unroll
    for (int e = 0; e < VC_ENS; e++) {
        const int ens = VC_ENS * qg + e;                                   // lanes 4 ens .. 4 ens + 3 of the group
        const bool have = 4 * ens < (int)Gd.count;
        const dabgpu_cw_desc Dd = descs[(size_t)Gd.first + (size_t)Gd.stride * (size_t)(have ? 4 * ens : 0)];
        on[e] = have && Dd.n_steps != 0;
        src[e] = (gptr)(uintptr_t)Dd.d_src;
        newest[e] = Dd.newest_slot;                                        // lane 4 ens is CIF 0: its newest slot is 4 nf
        // one ring geometry and one line phase per group, taken from the first ensemble that HAS a frame: a skipped ensemble
        // (ring decode, slot < 0) carries a zeroed descriptor
        if (on[e] && !any) {
            any = true; n_slots = Dd.n_slots; frame_stride = Dd.frame_stride; cif_stride = Dd.cif_stride;
            // memory lines: the class segments start at multiples of 64 bytes inside a row (cif_stride / 16 = 3456 = 54 x 64), the
            // sub-channel phi bytes into a line (4 bytes per capacity unit); line L of a stream = its bytes [64 L - phi, 64 L - phi + 64)
            phi = (int)((uintptr_t)Dd.d_src & 63);
        }
    }
    if (!any) return;                                                      // nothing to decode in this quarter: its lanes are never read
    // this thread's stream when loading: piece = (output CIF, class) = tid / 4, chunk tid % 4 of the line
    const int piece = tid >> 2, chunk = tid & 3, l_c16 = piece & 15;
    const int l_rel = (piece >> 4) - (15 - (int)(__brev((unsigned)l_c16) >> 28));       // output CIF - age
    size_t l_off[VC_ENS];                                                  // byte offset of the stream's line 0, chunk `chunk`, per ensemble
#pragma unroll
    for (int e = 0; e < VC_ENS; e++) {
        int slot = (int)newest[e] + l_rel;
        if (slot < 0) slot += (int)n_slots;
        if (slot >= (int)n_slots) slot -= (int)n_slots;
        l_off[e] = (size_t)(slot >> 2) * frame_stride + (size_t)(slot & 3) * cif_stride + (size_t)l_c16 * (cif_stride >> 4) + (size_t)(16 * chunk) - (size_t)phi;
    }
    unsigned char* const l_lds = ring + (piece >> 4) * VC_CIF_PITCH + l_c16 * VC_PITCH + 16 * chunk;   // + e * VC_ENS_PITCH + (line & 1) * 64

    const int real_rows = (n_in + 3) >> 2;
    const int n_tiles = (real_rows + VC_ROWS - 1) / VC_ROWS;
    const int last = (((n_in - 1) >> 4) + phi) >> 6;                       // nothing past the last line that holds input of this codeword
    // tile st reads class-stream bytes [64 st, 64 st + 64): lines st .. st + (phi != 0), clamped to `last`
    auto tile_hi = [&](int st) { return min(st + (phi ? 1 : 0), last); };
    u4v v[VC_ENS];
    auto fetch = [&](int line) {
#pragma unroll
        for (int e = 0; e < VC_ENS; e++) {
            v[e] = u4v{0u, 0u, 0u, 0u};
            if (on[e]) v[e] = *(const __attribute__((address_space(1))) u4v*)(src[e] + (l_off[e] + (size_t)(64 * line)));
        }
    };
    auto stash = [&](int line) {
#pragma unroll
        for (int e = 0; e < VC_ENS; e++) *reinterpret_cast<u4v*>(l_lds + e * VC_ENS_PITCH + (line & 1) * 64) = v[e];
    };
    int loaded = -1;                                                       // highest line in (or on its way to) the rings
    bool pend = false;                                                     // line `loaded` is in registers, not yet in LDS
    const int e = (tid >> 2) & 3, co = tid & 3;                            // when transposing: 16 adjacent threads = (ensemble, CIF)
    const uint32_t* const wb = reinterpret_cast<const uint32_t*>(ring) + e * (VC_ENS_PITCH / 4) + co * (VC_CIF_PITCH / 4);
    uint32_t* const obase = sym + Gd.sym_off + 16 * qg + 4 * e + co;
    for (int st = 0; st < n_tiles; st++) {
        if (pend) { stash(loaded); pend = false; }
        while (loaded < tile_hi(st)) { loaded++; fetch(loaded); stash(loaded); }          // the first tile (later ones: prefetched)
        __syncthreads();
        if (st + 1 < n_tiles && loaded < tile_hi(st + 1)) { loaded++; fetch(loaded); pend = true; }   // in flight while this tile is transposed
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int m = (tid >> 4) + 16 * it, q = m & 3, cb = m >> 2;   // class group q = row mod 4, columns 4 cb .. 4 cb + 3 of the tile
            const int col = (64 * st + 4 * cb + phi) & 127;               // phi is a multiple of 4: an aligned dword of the two-line ring
            const uint32_t* wp = wb + (4 * q) * (VC_PITCH / 4) + (col >> 2);
            const uint32_t A = wp[0], B = wp[VC_PITCH / 4], C = wp[2 * (VC_PITCH / 4)], D = wp[3 * (VC_PITCH / 4)];
            // 4 x 4 byte transpose: O[m'] = (A.b[m'], B.b[m'], C.b[m'], D.b[m']) = input bytes 4 j .. 4 j + 3 of row j = 4 (column) + q
            const uint32_t T0 = __builtin_amdgcn_perm(B, A, 0x05010400u), T1 = __builtin_amdgcn_perm(B, A, 0x07030602u);
            const uint32_t U0 = __builtin_amdgcn_perm(D, C, 0x05010400u), U1 = __builtin_amdgcn_perm(D, C, 0x07030602u);
            const uint32_t O[4] = {__builtin_amdgcn_perm(U0, T0, 0x05040100u), __builtin_amdgcn_perm(U0, T0, 0x07060302u),
                                   __builtin_amdgcn_perm(U1, T1, 0x05040100u), __builtin_amdgcn_perm(U1, T1, 0x07060302u)};
            const int jb = VC_ROWS * st + 16 * cb + q;
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                const int j = jb + 4 * mm;
                if (j < real_rows) obase[(size_t)j * 64] = vl_clamp4(O[mm]);
            }
        }
        __syncthreads();                                                   // the rings are rewritten by the next tile's stash
    }
}

// ---- group tables ----
// FIC: all codewords share one schedule; group g = codewords 64 g .. 64 g + 63
__global__ void vit_groups_uniform_kernel(dabgpu_vit_group* groups, size_t n_cw, uint32_t n_steps, uint32_t alloc_steps, uint32_t in_rows,
                                          uint32_t pi0, uint32_t st0, uint32_t pi1, uint32_t st1, uint32_t pi2, uint32_t st2,
                                          uint32_t pi3, uint32_t st3, dabgpu_vit_group_base base)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * 64 >= n_cw) return;
    dabgpu_vit_group G = {};
    G.first = base.first + (uint32_t)(g * 64); G.stride = 1;
    G.count = (uint32_t)((n_cw - g * 64 < 64) ? (n_cw - g * 64) : 64);
    G.n_steps = n_steps; G.alloc_steps = alloc_steps;
    G.seg_pi[0] = pi0; G.seg_steps[0] = st0; G.seg_pi[1] = pi1; G.seg_steps[1] = st1;
    G.seg_pi[2] = pi2; G.seg_steps[2] = st2; G.seg_pi[3] = pi3; G.seg_steps[3] = st3;
    G.in_rows = in_rows;
    G.sched_off = base.sched_off;
    G.sym_off = base.sym_off + g * (size_t)in_rows * 64;
    G.dec_off = base.dec_off + g * (size_t)alloc_steps * 128;
    G.res_delta = base.res_delta;
    groups[g] = G;
}

// MSC: codeword index = (4 e + c) * n_sub + s (msc_build_descs_kernel); group (li, gq) = the li-th lane-mapped sub-channel
// s = lane_subs[3 li] (decision-row prefix lane_subs[3 li + 1], symbol-row prefix lane_subs[3 li + 2]) of ensemble-CIFs 64 gq .. 64 gq + 63
__global__ void vit_groups_msc_kernel(dabgpu_vit_group* groups, const dabgpu_msc_plan* plans, const uint64_t* lane_subs,
                                      int n_lane_sub, int n_sub, size_t n_ens, uint32_t groups_per_sub, uint32_t sched_stride)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)n_lane_sub * groups_per_sub) return;
    const int li = (int)(g / groups_per_sub);
    const uint32_t gq = (uint32_t)(g - (size_t)li * groups_per_sub);
    const int s = (int)lane_subs[3 * li];
    const dabgpu_msc_plan P = plans[s];
    const size_t n_j = n_ens * 4;
    dabgpu_vit_group G = {};
    G.first = (uint32_t)((size_t)s + (size_t)n_sub * 64 * gq); G.stride = (uint32_t)n_sub;
    G.count = (uint32_t)((n_j - (size_t)gq * 64 < 64) ? (n_j - (size_t)gq * 64) : 64);
    G.n_steps = P.n_steps;
    G.alloc_steps = dabgpu_vit_alloc_steps(P.n_steps);
    for (int k = 0; k < 4; k++) { G.seg_pi[k] = P.seg_pi[k]; G.seg_steps[k] = P.seg_steps[k]; }
    G.in_rows = dabgpu_vit_in_rows(dabgpu_vit_in_bytes(P.seg_pi, P.seg_steps));
    G.sched_off = (uint64_t)li * sched_stride;
    G.sym_off = ((size_t)lane_subs[3 * li + 2] * groups_per_sub + (size_t)gq * G.in_rows) * 64;
    G.dec_off = ((size_t)lane_subs[3 * li + 1] * groups_per_sub + (size_t)gq * G.alloc_steps) * 128;
    groups[g] = G;
}

// schedule tables: one (row, selector) pair per trellis step, + the block the forward pass prefetches past the end
// grid (ceil(sched_stride / 256), n schedules)
__global__ void vit_sched_uniform_kernel(uint2* sched, uint32_t sched_stride, uint32_t pi0, uint32_t st0, uint32_t pi1, uint32_t st1,
                                         uint32_t pi2, uint32_t st2, uint32_t pi3, uint32_t st3, const dabgpu_vit_tables* tables)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sched_stride) return;
    const uint32_t seg_pi[4] = {pi0, pi1, pi2, pi3}, seg_steps[4] = {st0, st1, st2, st3};
    sched[t] = vl_sched_entry((int)t, seg_pi, seg_steps, dabgpu_vit_in_rows(dabgpu_vit_in_bytes(seg_pi, seg_steps)), tables);
}
__global__ void vit_sched_msc_kernel(uint2* sched, uint32_t sched_stride, const dabgpu_msc_plan* plans, const uint64_t* lane_subs,
                                     const dabgpu_vit_tables* tables)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, li = blockIdx.y;
    if (t >= sched_stride) return;
    const dabgpu_msc_plan P = plans[lane_subs[3 * li]];
    sched[(size_t)li * sched_stride + t] = vl_sched_entry((int)t, P.seg_pi, P.seg_steps, dabgpu_vit_in_rows(dabgpu_vit_in_bytes(P.seg_pi, P.seg_steps)), tables);
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_vit_groups_uniform_at(dabgpu_vit_group* d_groups, size_t n_cw, uint32_t n_steps, const uint32_t* seg_pi,
                                                          const uint32_t* seg_steps, dabgpu_vit_group_base base, hipStream_t stream)
{
    const size_t n_groups = (n_cw + 63) / 64;
    hipLaunchKernelGGL(dabgpu::vit_groups_uniform_kernel, dim3((unsigned)((n_groups + 127) / 128)), dim3(128), 0, stream,
                       d_groups, n_cw, n_steps, dabgpu_vit_alloc_steps(n_steps), dabgpu_vit_in_rows(dabgpu_vit_in_bytes(seg_pi, seg_steps)),
                       seg_pi[0], seg_steps[0], seg_pi[1], seg_steps[1],
                       seg_pi[2], seg_steps[2], seg_pi[3], seg_steps[3], base);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_vit_groups_uniform(dabgpu_vit_group* d_groups, size_t n_cw, uint32_t n_steps,
                                                       const uint32_t* seg_pi, const uint32_t* seg_steps, hipStream_t stream)
{
    return dabgpu_launch_vit_groups_uniform_at(d_groups, n_cw, n_steps, seg_pi, seg_steps, dabgpu_vit_group_base{0, 0, 0, 0, 0}, stream);
}

extern "C" hipError_t dabgpu_launch_vit_groups_msc(dabgpu_vit_group* d_groups, const dabgpu_msc_plan* d_plans,
                                                   const uint64_t* d_lane_subs, int n_lane_sub, int n_sub, size_t n_ens,
                                                   uint32_t groups_per_sub, uint32_t sched_stride, hipStream_t stream)
{
    const size_t n_groups = (size_t)n_lane_sub * groups_per_sub;
    hipLaunchKernelGGL(dabgpu::vit_groups_msc_kernel, dim3((unsigned)((n_groups + 127) / 128)), dim3(128), 0, stream,
                       d_groups, d_plans, d_lane_subs, n_lane_sub, n_sub, n_ens, groups_per_sub, sched_stride);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_vit_sched_uniform(uint2* d_sched, uint32_t sched_stride, const uint32_t* seg_pi, const uint32_t* seg_steps,
                                                      const dabgpu_vit_tables* d_tables, hipStream_t stream)
{
    hipLaunchKernelGGL(dabgpu::vit_sched_uniform_kernel, dim3((sched_stride + 255) / 256), dim3(256), 0, stream, d_sched, sched_stride,
                       seg_pi[0], seg_steps[0], seg_pi[1], seg_steps[1], seg_pi[2], seg_steps[2], seg_pi[3], seg_steps[3], d_tables);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_vit_sched_msc(uint2* d_sched, uint32_t sched_stride, const dabgpu_msc_plan* d_plans,
                                                  const uint64_t* d_lane_subs, int n_lane_sub, const dabgpu_vit_tables* d_tables, hipStream_t stream)
{
    hipLaunchKernelGGL(dabgpu::vit_sched_msc_kernel, dim3((sched_stride + 255) / 256, (unsigned)n_lane_sub), dim3(256), 0, stream, d_sched,
                       sched_stride, d_plans, d_lane_subs, d_tables);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_vit_prep(int kind, const dabgpu_vit_group* d_groups, size_t n_groups, uint32_t max_in_rows,
                                             const dabgpu_cw_desc* d_descs, uint32_t* d_sym, uint32_t groups_per_sub, hipStream_t stream)
{
    using namespace dabgpu;
    const unsigned tiles = (max_in_rows + VL_TILE - 1) / VL_TILE;
    if (kind == 3)      // direct, contiguous, 16-byte aligned codewords (FIC)
        hipLaunchKernelGGL(vit_prep_direct_kernel, dim3((unsigned)n_groups, tiles), dim3(256), 0, stream, d_groups, d_descs, d_sym);
    else if (kind == 2)      // ring of 4 CIFs per frame in class order
        if (groups_per_sub != 0 && n_groups % groups_per_sub == 0) {
            const unsigned n_lane_sub = (unsigned)(n_groups / groups_per_sub);
            hipLaunchKernelGGL(vit_prep_ring4c_kernel, dim3(8 * ((4 * groups_per_sub + 7) / 8) * n_lane_sub), dim3(256), 0, stream, d_groups, d_descs,
                               d_sym, groups_per_sub, n_lane_sub);
        } else {
            hipLaunchKernelGGL(vit_prep_ring4c_kernel, dim3((unsigned)(4 * n_groups)), dim3(256), 0, stream, d_groups, d_descs, d_sym, 0u, 1u);
        }
    else if (kind)
        hipLaunchKernelGGL(vit_prep_ring4_kernel, dim3((unsigned)n_groups, tiles), dim3(256), 0, stream, d_groups, d_descs, d_sym);
    else
        hipLaunchKernelGGL(vit_prep_kernel, dim3((unsigned)n_groups, tiles), dim3(256), 0, stream, d_groups, d_descs, d_sym);
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_vit_trellis(const dabgpu_vit_group* d_groups, size_t n_groups, const dabgpu_cw_desc* d_descs, uint32_t* d_sym,
                                                uint32_t* d_dec, dabgpu_cw_result* d_results, int tie_rule, const dabgpu_vit_tables* d_tables,
                                                const uint2* d_sched, int octet, int n_cu, hipStream_t stream)
{
    using namespace dabgpu;
    // eight lanes per codeword (viterbi_octet.hip) or one
    if (octet) return dabgpu_launch_viterbi_octet(d_groups, n_groups, d_descs, d_sym, d_dec, d_results, tie_rule, d_tables, d_sched, stream);
#define VL_GO(TIE, W) hipLaunchKernelGGL((vit_lanes_kernel<TIE, W>), dim3((unsigned)((n_groups + (W) - 1) / (W))), dim3(64 * (W)), 0, stream, \
                                         d_groups, (int)n_groups, d_descs, d_sym, d_dec, d_results, d_tables, d_sched)
    const bool four = n_groups > (size_t)n_cu && n_groups <= (size_t)4 * n_cu;
    const bool five_per_simd = n_groups > (size_t)16 * n_cu && n_groups <= (size_t)20 * n_cu;        // 4 SIMDs per CU
    if (five_per_simd) {
        if (tie_rule) hipLaunchKernelGGL((vit_lanes_kernel<1, 1, 5>), dim3((unsigned)n_groups), dim3(64), 0, stream, d_groups, (int)n_groups, d_descs, d_sym, d_dec, d_results, d_tables, d_sched);
        else hipLaunchKernelGGL((vit_lanes_kernel<0, 1, 5>), dim3((unsigned)n_groups), dim3(64), 0, stream, d_groups, (int)n_groups, d_descs, d_sym, d_dec, d_results, d_tables, d_sched);
    }
    else if (tie_rule) { if (four) VL_GO(1, 4); else VL_GO(1, 1); }
    else { if (four) VL_GO(0, 4); else VL_GO(0, 1); }
#undef VL_GO
    return hipGetLastError();
}

extern "C" hipError_t dabgpu_launch_viterbi_lanes(const dabgpu_vit_group* d_groups, size_t n_groups, uint32_t max_in_rows,
                                                  const dabgpu_cw_desc* d_descs, uint32_t* d_sym, uint32_t* d_dec,
                                                  dabgpu_cw_result* d_results, int tie_rule, int ring4, const dabgpu_vit_tables* d_tables,
                                                  const uint2* d_sched, int octet, int n_cu, uint32_t groups_per_sub, hipStream_t stream)
{
    hipError_t e = dabgpu_launch_vit_prep(ring4, d_groups, n_groups, max_in_rows, d_descs, d_sym, groups_per_sub, stream);
    if (e != hipSuccess) return e;
    return dabgpu_launch_vit_trellis(d_groups, n_groups, d_descs, d_sym, d_dec, d_results, tie_rule, d_tables, d_sched, octet, n_cu, stream);
}
