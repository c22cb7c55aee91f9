// pk16_rate.hip -- development microbenchmark: issue rate of packed 16-bit integer VALU ops on gfx950
// (v_pk_add_u16 / v_pk_sub_i16 / v_pk_min_u16) against v_add_u32, to size a two-codewords-per-wave Viterbi ACS.
// build: hipcc --offload-arch=gfx950 -O3 pk16_rate.hip -o pk16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
#define N_ACC 16
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a) {
    unsigned x[N_ACC];
    for (int i = 0; i < N_ACC; i++) x[i] = threadIdx.x * 77u + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
            if (MODE == 0) x[i] = x[i] + a;
            if (MODE == 1) { us2 v = __builtin_bit_cast(us2, x[i]); us2 w = __builtin_bit_cast(us2, a); v = v + w; x[i] = __builtin_bit_cast(unsigned, v); }
            if (MODE == 2) { s2 v = __builtin_bit_cast(s2, x[i]); s2 w = __builtin_bit_cast(s2, a); v = v - w; x[i] = __builtin_bit_cast(unsigned, v); }
            if (MODE == 3) { us2 v = __builtin_bit_cast(us2, x[i]); us2 w = __builtin_bit_cast(us2, a); v = __builtin_elementwise_min(v, w); x[i] = __builtin_bit_cast(unsigned, v) + 1u; }
            if (MODE == 4) x[i] = __builtin_amdgcn_sdot4((int)x[i], (int)a, (int)x[i], false);
            if (MODE == 5) x[i] = __builtin_amdgcn_alignbit(x[i], a, 31);
            if (MODE == 6) x[i] = __builtin_amdgcn_perm(x[i], a, 0x05040100u);
            if (MODE == 7) x[i] = __builtin_amdgcn_sad_u8(x[i], a, x[i]);
            if (MODE == 8) x[i] = __builtin_amdgcn_ubfe(x[i], a & 31, 9) + (x[i] << 3);
            if (MODE == 9) x[i] = (unsigned)__builtin_amdgcn_mov_dpp((int)x[i], 0x128, 0xF, 0xF, true) ^ a;
        }
    }
    unsigned s = 0; for (int i = 0; i < N_ACC; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    unsigned* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(unsigned) * 4);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0x00030005u); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 0x00030005u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * ITERS * N_ACC * (MODE == 3 ? 2 : 1) / 1024.0;
    printf("%-16s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {2, 4}) {
        run<0>("v_add_u32", w); run<1>("v_pk_add_u16", w); run<2>("v_pk_sub_i16", w); run<3>("v_pk_min_u16+add", w);
        run<4>("v_dot4_i32_i8", w); run<5>("v_alignbit", w); run<6>("v_perm_b32", w);
        run<7>("v_sad_u8", w); run<8>("v_bfe+lshl_add", w); run<9>("v_mov_dpp+xor", w);
    }
    return 0;
}
