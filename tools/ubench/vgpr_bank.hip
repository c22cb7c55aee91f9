// vgpr_bank.hip -- development microbenchmark: does a packed-integer VALU instruction whose two source VGPRs sit in the same register
// bank (register number mod 4) issue slower on gfx950?  Explicit registers: v_pk_add_u16 d, a, b with (a, b) in different / the same bank.
// build: hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 8192
template <int SAME>
__global__ __launch_bounds__(256) void k(unsigned* out) {
    unsigned acc = threadIdx.x;
    // 16 independent accumulators v32..v47; sources v16.. (bank pattern chosen below); everything in fixed registers
    asm volatile(
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n"
        "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"
        :: "v"(acc) : "v16","v17","v18","v19","v20","v21","v22","v23","v32","v33","v34","v35","v36","v37","v38","v39");
    for (int it = 0; it < ITERS; it++) {
        if (SAME == 0) {            // sources in DIFFERENT banks: (v32 b0, v17 b1), (v33 b1, v18 b2), ...
            asm volatile(
                "v_pk_add_u16 v32, v32, v17\n v_pk_add_u16 v33, v33, v18\n v_pk_add_u16 v34, v34, v19\n v_pk_add_u16 v35, v35, v16\n"
                "v_pk_add_u16 v36, v36, v21\n v_pk_add_u16 v37, v37, v22\n v_pk_add_u16 v38, v38, v23\n v_pk_add_u16 v39, v39, v20\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        } else if (SAME == 1) {     // sources in the SAME bank: (v32, v16), (v33, v17), ...
            asm volatile(
                "v_pk_add_u16 v32, v32, v16\n v_pk_add_u16 v33, v33, v17\n v_pk_add_u16 v34, v34, v18\n v_pk_add_u16 v35, v35, v19\n"
                "v_pk_add_u16 v36, v36, v20\n v_pk_add_u16 v37, v37, v21\n v_pk_add_u16 v38, v38, v22\n v_pk_add_u16 v39, v39, v23\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        } else if (SAME == 2) {     // v_pk_min_i16 d, a, b with d in a third register: different banks
            asm volatile(
                "v_pk_min_i16 v32, v33, v18\n v_pk_min_i16 v33, v34, v19\n v_pk_min_i16 v34, v35, v16\n v_pk_min_i16 v35, v36, v17\n"
                "v_pk_min_i16 v36, v37, v22\n v_pk_min_i16 v37, v38, v23\n v_pk_min_i16 v38, v39, v20\n v_pk_min_i16 v39, v32, v21\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        } else {                    // the same with both sources in one bank
            asm volatile(
                "v_pk_min_i16 v32, v33, v17\n v_pk_min_i16 v33, v34, v18\n v_pk_min_i16 v34, v35, v19\n v_pk_min_i16 v35, v36, v16\n"
                "v_pk_min_i16 v36, v37, v21\n v_pk_min_i16 v37, v38, v22\n v_pk_min_i16 v38, v39, v23\n v_pk_min_i16 v39, v32, v20\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        }
    }
    unsigned s;
    asm volatile("v_add_u32 %0, v32, v33\n v_add_u32 %0, %0, v34\n v_add_u32 %0, %0, v35\n v_add_u32 %0, %0, v36\n v_add_u32 %0, %0, v37\n v_add_u32 %0, %0, v38\n v_add_u32 %0, %0, v39"
                 : "=v"(s) :: "v32","v33","v34","v35","v36","v37","v38","v39");
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SAME> void run(const char* name, int waves_per_simd) {
    unsigned* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(unsigned) * 4);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SAME><<<blocks, 256>>>(d); hipDeviceSynchronize();
    hipEventRecord(e0); k<SAME><<<blocks, 256>>>(d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * ITERS * 8 / 1024.0;
    printf("%-44s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 4, 5}) {
        run<0>("v_pk_add_u16, sources in different banks", w); run<1>("v_pk_add_u16, sources in the same bank", w);
        run<2>("v_pk_min_i16, sources in different banks", w); run<3>("v_pk_min_i16, sources in the same bank", w);
    }
    return 0;
}
