// development microbenchmark: how fast ONE wavefront can issue VALU work, by dependency distance and by the number of
// wavefronts sharing the SIMD (gfx950).  ns per wave-instruction and SIMD; 2 cycles = ~0.95 ns at 2.1 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define R4(a) a a a a
#define R16(a) R4(a) R4(a) R4(a) R4(a)
// DIST = number of independent accumulators cycled through (1 = every instruction depends on the previous one)
template <int DIST> __global__ __launch_bounds__(256) void k(float* out, float a) {
    asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
                 "v_mov_b32 v56, 0x3f800347\n v_mov_b32 v60, 0x3f8020c5" ::: CLOB);
    for (int it = 0; it < ITERS; it++) {
        if constexpr (DIST == 1) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 2) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n v_fmac_f32_e32 v41, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 3) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n v_fmac_f32_e32 v41, v56, v60\n v_fmac_f32_e32 v42, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 4) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n v_fmac_f32_e32 v41, v56, v60\n v_fmac_f32_e32 v42, v56, v60\n v_fmac_f32_e32 v43, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 8) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n v_fmac_f32_e32 v41, v56, v60\n v_fmac_f32_e32 v42, v56, v60\n v_fmac_f32_e32 v43, v56, v60\n"
                                                       "v_fmac_f32_e32 v44, v56, v60\n v_fmac_f32_e32 v45, v56, v60\n v_fmac_f32_e32 v46, v56, v60\n v_fmac_f32_e32 v47, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 16) asm volatile(R4(R4("v_fmac_f32_e32 v40, v56, v60\n" "v_fmac_f32_e32 v41, v56, v60\n" "v_fmac_f32_e32 v42, v56, v60\n" "v_fmac_f32_e32 v43, v56, v60\n" "v_fmac_f32_e32 v44, v56, v60\n" "v_fmac_f32_e32 v45, v56, v60\n" "v_fmac_f32_e32 v46, v56, v60\n" "v_fmac_f32_e32 v47, v56, v60\n" "v_fmac_f32_e32 v48, v56, v60\n" "v_fmac_f32_e32 v49, v56, v60\n" "v_fmac_f32_e32 v50, v56, v60\n" "v_fmac_f32_e32 v51, v56, v60\n" "v_fmac_f32_e32 v52, v56, v60\n" "v_fmac_f32_e32 v53, v56, v60\n" "v_fmac_f32_e32 v54, v56, v60\n" "v_fmac_f32_e32 v55, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 12) asm volatile(R4(R4("v_fmac_f32_e32 v40, v56, v60\n" "v_fmac_f32_e32 v41, v56, v60\n" "v_fmac_f32_e32 v42, v56, v60\n" "v_fmac_f32_e32 v43, v56, v60\n" "v_fmac_f32_e32 v44, v56, v60\n" "v_fmac_f32_e32 v45, v56, v60\n" "v_fmac_f32_e32 v46, v56, v60\n" "v_fmac_f32_e32 v47, v56, v60\n" "v_fmac_f32_e32 v48, v56, v60\n" "v_fmac_f32_e32 v49, v56, v60\n" "v_fmac_f32_e32 v50, v56, v60\n" "v_fmac_f32_e32 v51, v56, v60\n")) ::: CLOB);
        if constexpr (DIST == 10) asm volatile(R4(R4("v_fmac_f32_e32 v40, v56, v60\n" "v_fmac_f32_e32 v41, v56, v60\n" "v_fmac_f32_e32 v42, v56, v60\n" "v_fmac_f32_e32 v43, v56, v60\n" "v_fmac_f32_e32 v44, v56, v60\n" "v_fmac_f32_e32 v45, v56, v60\n" "v_fmac_f32_e32 v46, v56, v60\n" "v_fmac_f32_e32 v47, v56, v60\n" "v_fmac_f32_e32 v48, v56, v60\n" "v_fmac_f32_e32 v49, v56, v60\n")) ::: CLOB);
    }
    float r; asm volatile("v_add_f32 %0, v40, v41" : "=v"(r) :: CLOB);
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFF] = r + a;
}
template <int DIST> void run(int waves_per_simd, int wg = 64) {
    float* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int blocks = 256 * 4 * waves_per_simd * 64 / wg;       // one-wave workgroups: the dispatcher deals them round-robin over the SIMDs
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<DIST><<<blocks, wg>>>(d, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; r++) { (void)hipEventRecord(e0); k<DIST><<<blocks, wg>>>(d, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double n = (double)ITERS * (DIST >= 10 ? 16 : 64) * DIST;
    printf("wg %d dist %d  waves/SIMD %d : %.3f ms  %.2f ns per wave-instr per wave, %.2f ns per instr per SIMD\n", wg, DIST, waves_per_simd, best, best * 1e6 / n, best * 1e6 / (n * waves_per_simd));
    (void)hipFree(d);
}
int main() {
    for (int w = 1; w <= 4; w++) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<8>(w); }
    run<1>(8); run<8>(8);
    for (int w = 1; w <= 4; w *= 2) { run<10>(w, 256); run<12>(w, 256); run<16>(w, 256); }
    for (int w = 1; w <= 4; w++) { run<1>(w, 256); run<2>(w, 256); run<4>(w, 256); run<8>(w, 256); }
    return 0;
}
