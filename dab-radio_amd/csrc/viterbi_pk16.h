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


}  // namespace dabgpu
