// viterbi_pk16.h -- packed-u16 helpers and compile-time trellis tables shared by the batch forms of the channel decoder
// (viterbi_lanes.hip: one lane per codeword; viterbi_octet.hip: eight lanes per codeword).  Device code, gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dabgpu {

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int VL_TILE = 64;                 // steps per prep tile
constexpr int VL_PRBS = 511;

__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ s2 add16(s2 a, s2 b) { return __builtin_bit_cast(s2, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)); }
// candidate = metric + branch cost under core model TIE: 0 = the scalar core's uint16_t sum, it wraps; 1 = the SIMD cores'
// adds_epu16, it saturates at 65535 -- on the biased representation (stored = value - 32768) that is the SIGNED saturating add of a
// non-negative cost: v_pk_add_i16 ... clamp, the same single instruction
template <int TIE>
__device__ __forceinline__ s2 addm(s2 a, s2 b) {
    if constexpr (TIE != 0) return __builtin_elementwise_add_sat(a, b);
    else return add16(a, b);
}
__device__ __forceinline__ s2 sub16(s2 a, s2 b) { return __builtin_bit_cast(s2, __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ s2 min16(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s2 satsub16(s2 a, s2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ s2 swap16(s2 v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ s2 lo_hi(s2 lo_src, s2 hi_src) { return __builtin_shufflevector(lo_src, hi_src, 0, 3); }

// (a & mask) | (b & ~mask) in one instruction (the compiler splits the expression into v_and + v_and_or)
__device__ __forceinline__ uint32_t vl_bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}

// ---- compile-time trellis tables ----
__host__ __device__ constexpr int vl_parity(unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return (int)(v & 1u); }
// sign pattern of butterfly b (input bit 0): bit 0 = polynomials 0 and 3 (109), bit 1 = polynomial 1 (79), bit 2 = polynomial 2 (83)
// expect +127 (dab_viterbi_decoder.cpp:25, ViterbiBranchTable)
__host__ __device__ constexpr int vl_sigma(int b) {
    return vl_parity((2u * (unsigned)b) & 109u) | (vl_parity((2u * (unsigned)b) & 79u) << 1) | (vl_parity((2u * (unsigned)b) & 83u) << 2);
}
__host__ __device__ constexpr int vl_ins_zero(int i, int q) { return ((i >> q) << (q + 1)) | (i & ((1 << q) - 1)); }
// pattern flip between the two butterflies of a register pair in phase q: b1 = b0 | 1 << q (q < 5); phase 5: lower vs upper predecessor
__host__ __device__ constexpr int vl_flip(int q) { return q < 5 ? (vl_sigma(0) ^ vl_sigma(1 << q)) : 7; }


// ---- branch costs of one trellis step: C[s] = (e(s), e(s ^ FLIP)) for the 8 sign patterns s ----
// e(s) = 508 - (+-a +- y1 +- y2), a = y0 + y3 (polynomials 0 and 3 are equal), bit k of s set = '+' for component k (a, y1, y2).
// FLIP = the pattern difference between the two metrics packed in one register (a property of the layout phase).  Split the
// components into the ones FLIP flips (F) and the others (U): e(s) = x - f, e(s ^ FLIP) = x + f with x = 508 - (U part), f = (F part).
// Every f and every x is ONE v_dot4_i32_i8 of the packed symbols with +-1 weights (x: accumulator 508), and a register is one
// v_pk_mad_i16 f.lo * (-1, +1) + x.lo (op_sel broadcasts): 4 registers hold all 8 patterns -- 8 or 9 instructions where pairing the
// values up with v_perm_b32 took 19.  C[s] of a pattern whose register holds it in the other order is that register with its halves
// swapped (op_sel of the consuming packed add, free).
template <int FLIP>
struct vl_costmap {
    static_assert(FLIP >= 1 && FLIP <= 7, "pattern difference of the two halves");
    static constexpr int lb = (FLIP & 1) ? 0 : ((FLIP & 2) ? 1 : 2);            // lowest flipped component: '+' in the canonical pattern
    static constexpr int o0 = lb == 0 ? 1 : 0, o1 = lb == 2 ? 1 : 2;            // the other two components
    static constexpr bool f0 = ((FLIP >> o0) & 1) != 0, f1 = ((FLIP >> o1) & 1) != 0;
    static constexpr int nF = 1 << ((f0 ? 1 : 0) + (f1 ? 1 : 0)), nX = 1 << ((f0 ? 0 : 1) + (f1 ? 0 : 1));
    // register of pattern s, and whether it holds (e(s ^ FLIP), e(s)) rather than (e(s), e(s ^ FLIP))
    __host__ __device__ static constexpr int idx_of(int s) { const int t = ((s >> lb) & 1) ? s : (s ^ FLIP); return ((t >> o0) & 1) | (((t >> o1) & 1) << 1); }
    __host__ __device__ static constexpr bool swapped(int s) { return ((s >> lb) & 1) == 0; }
    // which f / x value register idx uses
    __host__ __device__ static constexpr int fi(int idx) { return f0 ? ((idx & 1) | (f1 ? (idx & 2) : 0)) : (f1 ? ((idx >> 1) & 1) : 0); }
    __host__ __device__ static constexpr int xi(int idx) { return !f0 ? ((idx & 1) | (!f1 ? (idx & 2) : 0)) : (!f1 ? ((idx >> 1) & 1) : 0); }
    // v_dot4 weights (y0, y1, y2, y3) of f number j / x number j; sg* = +1 / -1: an extra sign per component (the lane's part of the
    // pattern in viterbi_octet.hip; all +1 in viterbi_lanes.hip)
    __host__ __device__ static constexpr int pack(int wa, int w1, int w2) { return (wa & 0xFF) | ((w1 & 0xFF) << 8) | ((w2 & 0xFF) << 16) | ((wa & 0xFF) << 24); }
    __host__ __device__ static constexpr int wF(int j, int sg0, int sg1, int sg2) {
        int w[3] = {0, 0, 0};
        const int sg[3] = {sg0, sg1, sg2};
        w[lb] = sg[lb];
        int n = 0;
        if (f0) { w[o0] = ((j >> n) & 1) ? sg[o0] : -sg[o0]; n++; }
        if (f1) { w[o1] = ((j >> n) & 1) ? sg[o1] : -sg[o1]; n++; }
        return pack(w[0], w[1], w[2]);
    }
    __host__ __device__ static constexpr int wX(int j, int sg0, int sg1, int sg2) {       // x = 508 - (U part): the weights carry the minus
        int w[3] = {0, 0, 0};
        const int sg[3] = {sg0, sg1, sg2};
        int n = 0;
        if (!f0) { w[o0] = ((j >> n) & 1) ? -sg[o0] : sg[o0]; n++; }
        if (!f1) { w[o1] = ((j >> n) & 1) ? -sg[o1] : sg[o1]; n++; }
        return pack(w[0], w[1], w[2]);
    }
};

__device__ __forceinline__ s2 vl_cost_reg(int f, int x) {          // (x - f, x + f) from the low halves
    uint32_t r;
    const uint32_t signs = 0x0001FFFFu;                             // (-1, +1)
    asm("v_pk_mad_i16 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(f), "s"(signs), "v"(x));
    return as_s2(r);
}

// The f and x values of a step: v_dot4_i32_i8 in its three-address form (the compiler prefers v_dot4c, which needs a v_mov of the
// accumulator in front of every one), all of a step's in ONE asm block that ends with the wait states gfx950 requires between a DOT
// instruction's VGPR write and a VALU read of it (no hardware interlock; the compiler pads its own dot products, it cannot see into
// an asm block).  WSGPR: the weights are compile-time constants kept in SGPRs (viterbi_lanes.hip) / per-lane VGPRs (viterbi_octet.hip).
// k508: a register holding 508 (loop-invariant in the caller).
template <int NF, int NX, bool WSGPR>
__device__ __forceinline__ void vl_cost_dots(uint32_t y, const int (&WF)[4], const int (&WX)[4], int k508, int (&fv)[4], int (&xv)[4]) {
    static_assert((NF == 1 && NX == 4) || (NF == 2 && NX == 2) || (NF == 4 && NX == 1), "vl_costmap");
    fv[0] = fv[1] = fv[2] = fv[3] = 0; xv[0] = xv[1] = xv[2] = xv[3] = k508;
    if constexpr (NF == 1) {
        if constexpr (WSGPR)
            asm("v_dot4_i32_i8 %0, %5, %6, 0\n\tv_dot4_i32_i8 %1, %5, %7, %11\n\tv_dot4_i32_i8 %2, %5, %8, %11\n\tv_dot4_i32_i8 %3, %5, %9, %11\n\t"
                "v_dot4_i32_i8 %4, %5, %10, %11\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
                : "v"(y), "s"(WF[0]), "s"(WX[0]), "s"(WX[1]), "s"(WX[2]), "s"(WX[3]), "v"(k508));
        else
            asm("v_dot4_i32_i8 %0, %5, %6, 0\n\tv_dot4_i32_i8 %1, %5, %7, %11\n\tv_dot4_i32_i8 %2, %5, %8, %11\n\tv_dot4_i32_i8 %3, %5, %9, %11\n\t"
                "v_dot4_i32_i8 %4, %5, %10, %11\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
                : "v"(y), "v"(WF[0]), "v"(WX[0]), "v"(WX[1]), "v"(WX[2]), "v"(WX[3]), "v"(k508));
    } else if constexpr (NF == 2) {
        if constexpr (WSGPR)
            asm("v_dot4_i32_i8 %0, %4, %5, 0\n\tv_dot4_i32_i8 %1, %4, %6, 0\n\tv_dot4_i32_i8 %2, %4, %7, %9\n\tv_dot4_i32_i8 %3, %4, %8, %9\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(fv[1]), "=&v"(xv[0]), "=&v"(xv[1])
                : "v"(y), "s"(WF[0]), "s"(WF[1]), "s"(WX[0]), "s"(WX[1]), "v"(k508));
        else
            asm("v_dot4_i32_i8 %0, %4, %5, 0\n\tv_dot4_i32_i8 %1, %4, %6, 0\n\tv_dot4_i32_i8 %2, %4, %7, %9\n\tv_dot4_i32_i8 %3, %4, %8, %9\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(fv[1]), "=&v"(xv[0]), "=&v"(xv[1])
                : "v"(y), "v"(WF[0]), "v"(WF[1]), "v"(WX[0]), "v"(WX[1]), "v"(k508));
    } else {
        if constexpr (WSGPR)
            asm("v_dot4_i32_i8 %0, %4, %5, 0\n\tv_dot4_i32_i8 %1, %4, %6, 0\n\tv_dot4_i32_i8 %2, %4, %7, 0\n\tv_dot4_i32_i8 %3, %4, %8, 0\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(fv[1]), "=&v"(fv[2]), "=&v"(fv[3])
                : "v"(y), "s"(WF[0]), "s"(WF[1]), "s"(WF[2]), "s"(WF[3]));
        else
            asm("v_dot4_i32_i8 %0, %4, %5, 0\n\tv_dot4_i32_i8 %1, %4, %6, 0\n\tv_dot4_i32_i8 %2, %4, %7, 0\n\tv_dot4_i32_i8 %3, %4, %8, 0\n\ts_nop 2"
                : "=&v"(fv[0]), "=&v"(fv[1]), "=&v"(fv[2]), "=&v"(fv[3])
                : "v"(y), "v"(WF[0]), "v"(WF[1]), "v"(WF[2]), "v"(WF[3]));
    }
}

template <int FLIP, bool WSGPR>
__device__ __forceinline__ void vl_cost_table(uint32_t ysym, const int (&WF)[4], const int (&WX)[4], int k508, s2 (&C)[8]) {
    using map = vl_costmap<FLIP>;
    int fv[4], xv[4];
    vl_cost_dots<map::nF, map::nX, WSGPR>(ysym, WF, WX, k508, fv, xv);
    s2 R[4];
#pragma unroll
    for (int i = 0; i < 4; i++) R[i] = vl_cost_reg(fv[map::fi(i)], xv[map::xi(i)]);
#pragma unroll
    for (int s = 0; s < 8; s++) C[s] = map::swapped(s) ? swap16(R[map::idx_of(s)]) : R[map::idx_of(s)];
}

}  // namespace dabgpu
