// viterbi_octet.hip -- gfx950 kernel of the MIDDLE form of the channel decoder: EIGHT LANES per codeword.
//
// viterbi.hip gives a codeword a whole wavefront (64 lanes = 64 states, ~33 issue slots per trellis step and codeword);
// viterbi_lanes.hip gives it one lane (all 64 states in 32 packed registers, ~2.9 VALU instructions per step and codeword, but a
// wavefront = 64 codewords is indivisible: a batch of G groups keeps min(G, 1024) SIMDs busy, and a lone wavefront on a SIMD issues at
// half rate).  Between them -- the FIB groups of a few thousand frames, the sub-channels of a few hundred ensembles -- sits this
// mapping: a wavefront decodes 8 codewords, a group of 64 codewords (one schedule, the same symbol array and gather kernels as the lane
// mapping) is a workgroup of 8 wavefronts, two per SIMD.
//   lane              = 8 l + c: codeword c of the wavefront, sub-lane l = 0..7
//   metrics           = 8 states per lane in 4 registers, two biased u16 per register (viterbi_lanes.hip): the 6 bits of a state's
//                       POSITION are (l: lane bits 5,4,3)(register: 2 bits)(half: 1 bit)
//   layout            = ROTATING.  A trellis step turns state s into 2 s + in: every state bit moves up one place, bit 5 leaves, a new
//                       bit 0 enters.  The position bits do not move: position bit ORD[q] is the one that holds state bit 5 in phase
//                       q = step mod 6 (the ACTIVE bit), the others hold state bits 4, 3, ... in the cyclic order of ORD; after the step
//                       the active position bit holds the new bit 0.  So the two predecessors (b, b + 32) of a butterfly always differ in
//                       the active position bit only, and the survivors (2 b, 2 b + 1) land on the same two positions:
//                         phases 0, 1  active = a register bit   -> in-lane, two register pairs, as viterbi_lanes.hip (8 ops per pair)
//                         phase 2      active = the half bit     -> in-lane, op_sel broadcasts (4 ops per register)
//                         phase 3      active = lane bit 3       -> the partner's candidate arrives by DPP row_ror:8
//                         phases 4, 5  active = lane bits 4, 5   -> v_permlane16_swap / v_permlane32_swap trade one candidate of each
//                                                                   lane for the partner's: afterwards one register holds every lane's
//                                                                   lower-predecessor candidate, the other the upper one
//   branch costs      = the pattern of a butterfly is linear in the state bits: (register bits) ^ (half bit) ^ (lane bits).  The lane's
//                       part is folded into the INPUTS -- the v_dot4 weights of the cost table (viterbi_pk16.h) carry per-lane,
//                       per-phase signs -- so the table of 8 costs is indexed by compile-time patterns only; in the swap phases the
//                       lanes that hold the upper predecessor flip all three signs (their table is the complement).
//   decisions         = sign bytes of the 4 difference registers (v_perm_b32 selectors 8..11), two v_bfi per step into a word of
//                       2 steps x 8 bits, stored as one u16 per lane and step pair: [pair][codeword][sub-lane], the same 512 bytes per
//                       step and group as the lane mapping's decision area
//   chain-back        = walks the POSITION: going back over a step replaces the step's active position bit by the decision -- all other
//                       bits stay; 16 bytes per step pair hold all 128 decisions.  The walk is a serial chain of dependent
//                       instructions, so the 8 lanes of a codeword SPLIT it: sub-lane l walks the l-th eighth of the steps, starting
//                       48 steps early from an arbitrary position -- survivors merge, so after the run-in it is on the true path with
//                       near certainty.  Certainty comes from the check: the position each lane arrives at must be the one the next
//                       lane assumed at its first own step, all the way down from sub-lane 0 (which starts from the true end state).
//                       A lane whose assumption was wrong walks its eighth again from the true position, and the check repeats
//                       (exact after at most 8 rounds; one round for any signal whose survivors merge within the run-in).
// Arithmetic, tie rules, renormalisation and outputs are those of viterbi.hip / viterbi_lanes.hip / oracle/dab_oracle_decode.c
// (bit-exact, incl. path_error).  Reference: dab_viterbi_decoder.cpp:114-181, fic_decoder.cpp:53-117.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "dabgpu_internal.h"
#include "viterbi_pk16.h"

namespace dabgpu {

// position bits: H = 0 (half), R0 = 1, R1 = 2 (register = R1 R0), L3 = 3, L4 = 4, L5 = 5 (sub-lane l = position >> 3)
// ORD[q] = the position bit that is active (holds state bit 5) in phase q
__host__ __device__ constexpr int vo_ord(int q) { return q == 0 ? 1 : (q == 1 ? 2 : (q == 2 ? 0 : q)); }
__host__ __device__ constexpr int vo_ord_of(int slot) { return slot == 1 ? 0 : (slot == 2 ? 1 : (slot == 0 ? 2 : slot)); }
// state bit held by position bit `slot` in phase q (5 = active)
__host__ __device__ constexpr int vo_jbit(int slot, int q) { return (q + 5 - vo_ord_of(slot)) % 6; }
// contribution of a set position bit to the butterfly's sign pattern (0 for the active bit: it is not part of the butterfly index)
__host__ __device__ constexpr int vo_slot_sig(int slot, int q) { return vo_jbit(slot, q) < 5 ? vl_sigma(1 << vo_jbit(slot, q)) : 0; }
__host__ __device__ constexpr int vo_reg_pat(int r, int q) { return ((r & 1) ? vo_slot_sig(1, q) : 0) ^ ((r & 2) ? vo_slot_sig(2, q) : 0); }
__host__ __device__ constexpr int vo_flip_h(int q) { return vo_ord(q) == 0 ? 7 : vo_slot_sig(0, q); }
// bit of the chain-back's position word P that stands for a position bit: P = [L5:6][L4:5][L3:4][H:3][R0:2][step parity:1][R1:0],
// so that (P >> 5) picks one of the 4 decision dwords of a step pair and P & 31 the bit inside it (decision layout below)
__host__ __device__ constexpr int vo_pbit(int slot) { return slot == 0 ? 3 : (slot == 1 ? 2 : (slot == 2 ? 0 : slot + 1)); }

struct vo_lane_consts {
    int wf[6][4], wx[6][4];           // v_dot4 weights per phase (vl_costmap): the lane's part of the sign pattern folded in
    uint32_t sgn3;                    // +1 / -1 packed: orientation of the difference in phase 3 (DPP exchange)
    int k508;                         // 508 in a register: accumulator constant of the cost dot products
};

// saturating a * b (b = +-1): the difference register with the orientation the lane needs
__device__ __forceinline__ s2 vo_satmul(s2 a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(r) : "v"(as_u32(a)), "v"(b));
    return as_s2(r);
}

// one trellis step in phase Q, in place; g0 / g1 = sign bytes of the decisions of registers (0, 1) / (2, 3):
// byte 2 R0 + H of g<R1> = 0xFF iff the survivor of that position came from the upper predecessor (TIE 1: the complement)
template <int Q, int TIE>
__device__ __forceinline__ void vo_step(s2 (&M)[4], uint32_t ysym, const vo_lane_consts& K, uint32_t& g0, uint32_t& g1) {
    s2 C[8], D[4];
    vl_cost_table<vo_flip_h(Q), false>(ysym, K.wf[Q], K.wx[Q], K.k508, C);
    if constexpr (Q < 2) {                                     // active = register bit R0 (Q = 0) / R1 (Q = 1)
        constexpr int rb = Q == 0 ? 1 : 2;
#pragma unroll
        for (int ra = 0; ra < 4; ra++) {
            if (ra & rb) continue;
            const int rbb = ra | rb, s = vo_reg_pat(ra, Q);
            const s2 c1 = C[s], c2 = C[7 - s];
            const s2 t1 = addm<TIE>(M[ra], c1), t2 = addm<TIE>(M[rbb], c2);           // new state 2b:   lower + e | upper + (1016 - e)
            const s2 t3 = addm<TIE>(M[ra], c2), t4 = addm<TIE>(M[rbb], c1);           // new state 2b+1: lower + (1016 - e) | upper + e
            M[ra] = min16(t1, t2);
            M[rbb] = min16(t3, t4);
            D[ra] = TIE ? satsub16(t1, t2) : satsub16(t2, t1);
            D[rbb] = TIE ? satsub16(t3, t4) : satsub16(t4, t3);
        }
    } else if constexpr (Q == 2) {                             // active = the half bit: M[r] = (old[b], old[b + 32]), C[s] = (e, 1016 - e)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int s = vo_reg_pat(r, Q);
            const s2 lower = addm<TIE>(__builtin_shufflevector(M[r], M[r], 0, 0), C[s]);
            const s2 upper = addm<TIE>(__builtin_shufflevector(M[r], M[r], 1, 1), C[7 - s]);
            M[r] = min16(lower, upper);
            D[r] = TIE ? satsub16(lower, upper) : satsub16(upper, lower);
        }
    } else if constexpr (Q == 3) {                             // active = lane bit 3: partner = lane ^ 8, DPP row_ror:8
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int s = vo_reg_pat(r, Q);
            const s2 own = addm<TIE>(M[r], C[s]);                                  // this lane's predecessor -> this lane's new state (cost e)
            const s2 give = addm<TIE>(M[r], C[7 - s]);                             // this lane's predecessor -> the partner's new state (1016 - e)
            const s2 recv = as_s2((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(give), 0x128, 0xF, 0xF, true));
            M[r] = min16(own, recv);
            // lanes with bit 3 clear hold the LOWER predecessor: upper - lower = recv - own; the others: own - recv = -(recv - own).
            // TIE 1 wants lower - upper.  One saturating multiply by +-1 (K.sgn3) gives every lane the difference it needs
            D[r] = vo_satmul(satsub16(recv, own), K.sgn3);
        }
    } else {                                                   // active = lane bit 4 / 5: swap one candidate with the partner
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int s = vo_reg_pat(r, Q);
            // (the lanes that hold the upper predecessor use the complemented table, folded into their weights: their x is the
            // candidate for the PARTNER's new state, their y the one for their own)
            const uint32_t x = as_u32(addm<TIE>(M[r], C[s])), y = as_u32(addm<TIE>(M[r], C[7 - s]));
            s2 lo, up;
            if constexpr (Q == 4) { const auto v = __builtin_amdgcn_permlane16_swap(x, y, false, false); lo = as_s2(v[0]); up = as_s2(v[1]); }
            else { const auto v = __builtin_amdgcn_permlane32_swap(x, y, false, false); lo = as_s2(v[0]); up = as_s2(v[1]); }
            M[r] = min16(lo, up);                                              // lo / up: the candidate through the lower / upper predecessor
            D[r] = TIE ? satsub16(lo, up) : satsub16(up, lo);
        }
    }
    g0 = __builtin_amdgcn_perm(as_u32(D[1]), as_u32(D[0]), 0x0B0A0908u);
    g1 = __builtin_amdgcn_perm(as_u32(D[3]), as_u32(D[2]), 0x0B0A0908u);
}

// value of `v` in lane ^ (8 << k), k = 0, 1, 2
template <int KB>
__device__ __forceinline__ uint32_t vo_xchg(uint32_t v, int lane) {
    if constexpr (KB == 0) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);
    else if constexpr (KB == 1) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
}

// the reference's renormalisation (dab_viterbi_decoder.cpp:31-41): `hit` = this lane's codeword has metric[0] at the threshold
__device__ __forceinline__ void vo_renorm(s2 (&M)[4], uint32_t& total, bool hit, int lane) {
    s2 mn = min16(min16(M[0], M[1]), min16(M[2], M[3]));
    mn = min16(mn, swap16(mn));
    mn = min16(mn, as_s2(vo_xchg<0>(as_u32(mn), lane)));
    mn = min16(mn, as_s2(vo_xchg<1>(as_u32(mn), lane)));
    mn = min16(mn, as_s2(vo_xchg<2>(as_u32(mn), lane)));
    if (hit) {
        const uint32_t mu = ((uint32_t)(uint16_t)mn.x) ^ 0x8000u;           // unbiased minimum of the 64 states
        const s2 sub = as_s2(mu | (mu << 16));
#pragma unroll
        for (int r = 0; r < 4; r++) M[r] = sub16(M[r], sub);
        total += mu;
    }
}

constexpr int VO_OBYTES = 128;                                     // codewords up to this many output bytes keep them in LDS for the CRC

// workgroup = 8 wavefronts = one group of 64 codewords; wavefront w decodes codewords 8 w .. 8 w + 7
template <int TIE>
__global__ __launch_bounds__(512)
void vit_octet_kernel(const dabgpu_vit_group* __restrict__ groups, const dabgpu_cw_desc* __restrict__ descs,
                      const uint32_t* __restrict__ sym, uint32_t* __restrict__ dec, dabgpu_cw_result* __restrict__ results,
                      const dabgpu_vit_tables* __restrict__ tables, const uint2* __restrict__ sched, int warm_chunks)
{
    __shared__ unsigned char prbs[512];
    __shared__ unsigned short crc_tab[256];                        // CRC16 (x^16 + x^12 + x^5 + 1), one byte at a time
    __shared__ unsigned char obuf[8][8][VO_OBYTES];                // decoded bytes of CRC-checked codewords (FIB groups): [wavefront][codeword]
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 512; e += 512) prbs[e] = tables->prbs[e];
    if (threadIdx.x < 256) {
        unsigned v = threadIdx.x << 8;
        for (int qq = 0; qq < 8; qq++) v = (v & 0x8000u) ? (((v << 1) ^ 0x1021u) & 0xFFFFu) : ((v << 1) & 0xFFFFu);
        crc_tab[threadIdx.x] = (unsigned short)v;
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const dabgpu_vit_group Gd = groups[blockIdx.x];
    if (8 * wv >= (int)Gd.count) return;
    const int T = (int)Gd.n_steps;
    const int c = lane & 7, l = lane >> 3;
    const int cg = 8 * wv + c;                                     // codeword inside the group
    const bool valid = cg < (int)Gd.count;
    const size_t cw = (size_t)Gd.first + (size_t)Gd.stride * (size_t)(valid ? cg : 0);
    const dabgpu_cw_desc Dd = descs[cw];
    const bool live = valid && Dd.n_steps != 0;                    // n_steps == 0: skipped work item of a ring decode

    // ---- per-lane constants: the lane's part of the sign pattern, as v_dot4 weights ----
    vo_lane_consts K;
    auto lane_weights = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        using map = vl_costmap<vo_flip_h(q)>;
        int ls = ((l & 1) ? vo_slot_sig(3, q) : 0) ^ ((l & 2) ? vo_slot_sig(4, q) : 0) ^ ((l & 4) ? vo_slot_sig(5, q) : 0);
        if (q >= 4 && ((l >> (q - 3)) & 1)) ls ^= 7;               // swap phases: upper-predecessor lanes use the complemented table
        const int s0 = (ls & 1) ? -1 : 1, s1 = (ls & 2) ? -1 : 1, s2_ = (ls & 4) ? -1 : 1;
#pragma unroll
        for (int jn = 0; jn < 4; jn++) { K.wf[q][jn] = map::wF(jn, s0, s1, s2_); K.wx[q][jn] = map::wX(jn, s0, s1, s2_); }
    };
    lane_weights(std::integral_constant<int, 0>{}); lane_weights(std::integral_constant<int, 1>{}); lane_weights(std::integral_constant<int, 2>{});
    lane_weights(std::integral_constant<int, 3>{}); lane_weights(std::integral_constant<int, 4>{}); lane_weights(std::integral_constant<int, 5>{});
    K.k508 = 508;
    asm volatile("" : "+v"(K.k508));
    {
        const bool upper3 = (l & 1) != 0;
        K.sgn3 = ((TIE ? !upper3 : upper3) ? 0xFFFFFFFFu : 0x00010001u);
    }

    const uint32_t* grp_sym = sym + Gd.sym_off;                    // [row][64 codewords of the group]
    unsigned short* grp_dec = reinterpret_cast<unsigned short*>(dec + Gd.dec_off);   // [pair][64][8] u16
    const uint2* sch = sched + Gd.sched_off;
    const uint32_t cg4 = 4u * (uint32_t)cg, dec2 = 2u * (uint32_t)(8 * cg + l);

    // ---- forward pass: the LAST step runs in phase 5 (n_steps = 8 m + 6 is even: the first, partial block starts in phase 0, 2 or 4) ----
    const int q0 = (6 - T % 6) % 6;
    s2 M[4];
    {
        // start state ss in the layout of phase q0: state bit j sits in position bit ORD[(q0 + 5 - j) % 6]
        const uint32_t ss = Dd.start_state & 63u;
        uint32_t p0 = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int k0 = (5 - j + 6) % 6, k2 = (2 + 5 - j) % 6, k4 = (4 + 5 - j) % 6;       // q0 = 0, 2, 4
            const int slot = q0 == 0 ? vo_ord(k0) : (q0 == 2 ? vo_ord(k2) : vo_ord(k4));
            p0 |= ((ss >> j) & 1u) << slot;
        }
        const uint32_t non = (5080u ^ 0x8000u), st = 0x8000u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool mine = (p0 >> 3) == (uint32_t)l && ((p0 >> 1) & 3u) == (uint32_t)r;
            const uint32_t lo = (mine && (p0 & 1u) == 0) ? st : non, hi = (mine && (p0 & 1u) == 1) ? st : non;
            M[r] = as_s2(lo | (hi << 16));
        }
    }
    uint32_t total = 0;
    // two trellis steps t, t + 1 (t even) in phases Q, Q + 1; decisions of the pair: bit 8 H + 4 R0 + 2 (t & 1) + R1 of a u16 per lane
#define VO_PAIR(Q, TT, YA, YB)                                                                        \
    {                                                                                                 \
        uint32_t ga0, ga1, gb0, gb1;                                                                  \
        vo_step<Q, TIE>(M, YA, K, ga0, ga1);                                                          \
        vo_check_renorm();                                                                            \
        vo_step<(Q) + 1, TIE>(M, YB, K, gb0, gb1);                                                    \
        vo_check_renorm();                                                                            \
        uint32_t acc = ga0 & 0x01010101u;                                                             \
        acc = vl_bfi(0x02020202u, ga1, acc);                                                          \
        acc = vl_bfi(0x04040404u, gb0, acc);                                                          \
        acc = vl_bfi(0x08080808u, gb1, acc);                                                          \
        acc &= 0x0F0F0F0Fu;                                                                           \
        acc |= acc >> 12;                                                                             \
        if (TIE) acc = ~acc;                                                                          \
        *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(grp_dec + (size_t)((TT) >> 1) * 512) + dec2) = (unsigned short)acc; \
    }
    // metric[0] = position 0 = sub-lane 0, register 0, low half: the lanes 0..7 of the wavefront hold it for their codewords
    auto vo_check_renorm = [&]() {
        const uint64_t m = __ballot((int)M[0].x >= (int)(60455 - 32768)) & 0xFFull;
        if (m != 0) vo_renorm(M, total, ((m >> c) & 1ull) != 0, lane);
    };
#define VO_ROWS(LO, HI, SEL, TT)                                                                      \
    {                                                                                                 \
        const uint2 e_ = sch[TT];                                                                     \
        const char* rp_ = reinterpret_cast<const char*>(grp_sym + (size_t)e_.x * 64);                 \
        LO = *reinterpret_cast<const uint32_t*>(rp_ + cg4);                                           \
        HI = *reinterpret_cast<const uint32_t*>(rp_ + 256 + cg4);                                     \
        SEL = e_.y;                                                                                   \
    }
    int t = 0;
    if (q0 == 2) {
        uint32_t a0, a1, b0, b1, sa, sb;
        VO_ROWS(a0, a1, sa, t) VO_ROWS(b0, b1, sb, t + 1)
        VO_PAIR(2, t, __builtin_amdgcn_perm(a1, a0, sa), __builtin_amdgcn_perm(b1, b0, sb))
        t += 2;
    }
    if (q0 != 0) {
        uint32_t a0, a1, b0, b1, sa, sb;
        VO_ROWS(a0, a1, sa, t) VO_ROWS(b0, b1, sb, t + 1)
        VO_PAIR(4, t, __builtin_amdgcn_perm(a1, a0, sa), __builtin_amdgcn_perm(b1, b0, sb))
        t += 2;
    }
    uint32_t nlo[6], nhi[6], nsel[6];
#pragma unroll
    for (int q = 0; q < 6; q++) VO_ROWS(nlo[q], nhi[q], nsel[q], t + q)
    for (; t < T; t += 6) {
        uint32_t y[6];
#pragma unroll
        for (int q = 0; q < 6; q++) y[q] = __builtin_amdgcn_perm(nhi[q], nlo[q], nsel[q]);
#pragma unroll
        for (int q = 0; q < 6; q++) VO_ROWS(nlo[q], nhi[q], nsel[q], t + 6 + q)               // one block ahead (the table runs a block past the end)
        VO_PAIR(0, t, y[0], y[1])
        VO_PAIR(2, t + 2, y[2], y[3])
        VO_PAIR(4, t + 4, y[4], y[5])
    }
#undef VO_ROWS
#undef VO_PAIR

    // ---- end metric: the layout after the last step is that of phase 0 ----
    const uint32_t es = Dd.end_state & 63u;
    uint32_t pe = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) pe |= ((es >> j) & 1u) << vo_ord((5 - j + 6) % 6);
    uint32_t endm = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) endm = (((pe >> 1) & 3u) == (uint32_t)r) ? as_u32(M[r]) : endm;
    endm = (((pe & 1u) ? (endm >> 16) : endm) & 0xFFFFu) ^ 0x8000u;
    endm = (uint32_t)__shfl((int)endm, (int)(8u * (pe >> 3) + (uint32_t)c));          // from the sub-lane that holds the end state

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);               // decision words are re-read by other lanes of this wavefront: drain the stores first
                                                 // (same CU, same L2: no write-back is needed, an agent-scope release cost 60 us per launch)

    // ---- chain-back over steps T-1 .. 6 (dab_viterbi_decoder.cpp:124-129): decoded bit t-6 = decision of the survivor at step t ----
    const bool raw = (Dd.flags & DABGPU_CW_RAW) != 0;
    unsigned char* out = reinterpret_cast<unsigned char*>(Dd.d_out);
    const int n_out = (T - 6) >> 3;
    const bool crc_lds = Dd.n_crc_blocks != 0 && n_out <= VO_OBYTES;
    unsigned char* const ob = &obuf[wv][c][0];
    // position word P (vo_pbit) of the end state; the step parity bit is 1 at the top of every chunk (its first step is odd)
    uint32_t P_end = 2u;
#pragma unroll
    for (int slot = 0; slot < 6; slot++) P_end |= ((pe >> slot) & 1u) << vo_pbit(slot);
    // chunks of 24 steps = 12 sixteen-byte loads = 3 output bytes: every chunk starts in the layout of phase 0 at bit 7 of a byte, so
    // phases, parities and byte boundaries are compile-time.  Chunk i covers steps T-1-24i-23 .. T-1-24i.
    constexpr int CB = 24;
    const int n_chunks = (n_out + 2) / 3;
    const int seg = (n_chunks + 7) >> 3;                           // chunks per sub-lane
    const uint32_t* dec_cw = reinterpret_cast<const uint32_t*>(grp_dec) + 4 * cg;      // + pair * 256 dwords
    // Round 0: sub-lane l walks chunks [l seg, (l + 1) seg) after a run-in of warm_chunks chunks from an arbitrary position.
    // Then the links are checked: the position sub-lane l - 1 arrived at is the true position at sub-lane l's first own step (by
    // induction from sub-lane 0, which starts from the end state).  A lane that assumed another one walks its chunks again from the
    // true position, and the check repeats -- after round k the sub-lanes 0..k are certainly right, so at most 8 rounds; with merged
    // survivors (every signal worth decoding) round 0 is the only one.
    const int i_own = min(l * seg, n_chunks), i_end = min((l + 1) * seg, n_chunks);    // this lane's chunks
    uint32_t P_own = 0, P_exit = P_end;                            // position assumed at the first own step / reached after the last
    bool redo = true;                                              // this lane walks in the current round
    for (int round = 0; round < 9; round++) {
        const int i_beg = round == 0 ? max(i_own - warm_chunks, 0) : i_own;
        // (a run-in that reaches the top starts from the end state itself and is exact)
        uint32_t P = round == 0 ? (i_beg == 0 ? P_end : 2u) : P_own;
        const int trip = __builtin_amdgcn_readfirstlane(min(seg + (round == 0 ? warm_chunks : 0), n_chunks));   // longest walk of the wavefront
        for (int it = 0; it < trip; it++) {
            const int i = i_beg + it;
            if (round == 0 && i == i_own) P_own = P;
            const bool act = redo && i < i_end;
            const int th = T - 1 - CB * (act ? i : 0);
            u4v W[CB / 2];
#pragma unroll
            for (int v = 0; v < CB / 2; v++) {
                const int pr = (th >> 1) - v;
                W[v] = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(dec_cw + (size_t)(pr < 0 ? 0 : pr) * 256));
            }
            uint32_t Pn = P;
#pragma unroll
            for (int ub = 0; ub < CB; ub += 8) {
                uint32_t acc = 0;
#pragma unroll
                for (int u = ub; u < ub + 8; u++) {
                    // step th - u (th odd): pair (th >> 1) - u / 2, parity 1 - (u & 1), phase (5 - u) mod 6
                    const u4v w4 = W[u >> 1];
                    const uint32_t wlo = (Pn & 32u) ? w4.y : w4.x, whi = (Pn & 32u) ? w4.w : w4.z;
                    const uint32_t word = (Pn & 64u) ? whi : wlo;
                    const uint32_t d = (word >> (Pn & 31u)) & 1u;
                    const int q = ((5 - u) % 6 + 6) % 6;
                    const int sb = vo_pbit(vo_ord(q));
                    const uint32_t nextpar = (uint32_t)(u & 1) << 1;       // parity of step th - u - 1
                    Pn = (Pn & ~((1u << sb) | 2u)) | (d << sb) | nextpar;
                    acc |= d << (u - ub);                                  // bit index (t - 6) & 7 = 7 - (u - ub), MSB first
                }
                const int k = (th - ub - 7 - 6) >> 3;                      // byte of bits t-6 for t = th-ub-7 .. th-ub
                if (act && th - ub >= 6 && i >= i_own) {                   // (steps below 6 exist only in the last chunk's tail)
                    const unsigned char pb = raw ? (unsigned char)0 : prbs[k % VL_PRBS];
                    const unsigned char byte = (unsigned char)(acc ^ pb);
                    if (live) out[k] = byte;
                    if (crc_lds) ob[k] = byte;
                }
            }
            if (act) P = Pn;
        }
        if (redo) P_exit = P;
        // the position sub-lane l - 1 arrived at = the position at step T-1-24 (l seg), this lane's first own step
        const uint32_t P_prev = (uint32_t)__shfl((int)P_exit, (lane - 8) & 63);
        redo = l != 0 && i_own < n_chunks && P_own != P_prev;
        if (__ballot(redo) == 0) break;
        P_own = P_prev;
    }

    // ---- optional FIB CRC16 (fic_decoder.cpp:19-31,103-116): sub-lane l checks blocks l, l + 8, ... of its codeword ----
    uint32_t crc_mask = 0;
    if (__ballot(live && Dd.n_crc_blocks != 0)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                             // the bytes were written by other lanes of this wavefront
        if (live && Dd.n_crc_blocks) {
            const int blk_bytes = n_out / (int)Dd.n_crc_blocks;
            for (int b = l; b < (int)Dd.n_crc_blocks && b < 32; b += 8) {
                unsigned crc = 0xFFFFu;
                unsigned rx;
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
        crc_mask |= vo_xchg<0>(crc_mask, lane);
        crc_mask |= vo_xchg<1>(crc_mask, lane);
        crc_mask |= vo_xchg<2>(crc_mask, lane);
    }
    if (valid && l == 0) {
        dabgpu_cw_result R;
        R.path_error = live ? (uint64_t)total + endm : 0;
        R.crc_ok_mask = crc_mask;
        R.n_out_bytes = live ? (uint32_t)n_out : 0u;
        *reinterpret_cast<dabgpu_cw_result*>(reinterpret_cast<char*>(results + cw) + Gd.res_delta) = R;
    }
}

}  // namespace dabgpu

extern "C" hipError_t dabgpu_launch_viterbi_octet(const dabgpu_vit_group* d_groups, size_t n_groups, const dabgpu_cw_desc* d_descs,
                                                  const uint32_t* d_sym, uint32_t* d_dec, dabgpu_cw_result* d_results, int tie_rule,
                                                  const dabgpu_vit_tables* d_tables, const uint2* d_sched, hipStream_t stream)
{
    using namespace dabgpu;
    // run-in of the split chain-back in chunks of 24 steps: 2 = 48 steps (the survivors of this K = 7 code have merged long before).
    // DABGPU_VIT_OCTET_WARM=0 makes nearly every link fail: the tests use it to drive the serial second pass
    int warm = 2;
    if (const char* e = getenv("DABGPU_VIT_OCTET_WARM")) { const int v = atoi(e); if (v >= 0 && v <= 64) warm = v; }
    if (tie_rule) hipLaunchKernelGGL(vit_octet_kernel<1>, dim3((unsigned)n_groups), dim3(512), 0, stream, d_groups, d_descs, d_sym, d_dec, d_results, d_tables, d_sched, warm);
    else hipLaunchKernelGGL(vit_octet_kernel<0>, dim3((unsigned)n_groups), dim3(512), 0, stream, d_groups, d_descs, d_sym, d_dec, d_results, d_tables, d_sched, warm);
    return hipGetLastError();
}
