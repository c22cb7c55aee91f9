// development microbenchmark: source-operand patterns of v_fmac_f32 on gfx950 (follow-up of wave_issue.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define R4(a) a a a a
#define R16(a) R4(a) R4(a) R4(a) R4(a)
#define INIT "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" \
             "v_mov_b32 v56, 0x3f800347\n v_mov_b32 v57, 0x3f800347\n v_mov_b32 v60, 0x3f8020c5\n v_mov_b32 v61, 0x3f8020c5\n v_mov_b32 v62, 0x3f8020c5\n"
template <int T> __global__ __launch_bounds__(256) void k(float* out, float a) {
    asm volatile(INIT ::: CLOB);
    for (int it = 0; it < ITERS; it++) {
        if constexpr (T == 0) asm volatile(R16(R4("v_fmac_f32_e32 v40, v56, v60\n")) ::: CLOB);      // same register pair, both bank 0
        if constexpr (T == 1) asm volatile(R16(R4("v_fmac_f32_e32 v40, v60, v61\n")) ::: CLOB);      // banks 0 and 1
        if constexpr (T == 2) asm volatile(R16(R4("v_fmac_f32_e32 v40, v57, v62\n")) ::: CLOB);      // banks 1 and 2, dst bank 0
        if constexpr (T == 3) asm volatile(R16(R4("v_fmac_f32_e32 v41, v56, v60\n")) ::: CLOB);      // srcs bank 0, dst bank 1
        if constexpr (T == 4) asm volatile(R16(R4("v_mul_f32_e32 v40, v56, v60\n")) ::: CLOB);       // no accumulator read
        if constexpr (T == 5) asm volatile(R16(R4("v_mul_f32_e32 v40, v60, v61\n")) ::: CLOB);
        if constexpr (T == 6) asm volatile(R16(R4("v_add_f32_e32 v40, v56, v40\n")) ::: CLOB);       // dependent adds
        if constexpr (T == 7) asm volatile(R16(R4("v_add_f32_e32 v40, v57, v40\n")) ::: CLOB);
    }
    float r; asm volatile("v_add_f32 %0, v40, v41" : "=v"(r) :: CLOB);
    out[(blockIdx.x * 256 + threadIdx.x) & 0xFFFF] = r + a;
}
template <int T> void run(int w) {
    float* d; (void)hipMalloc(&d, 65536 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<T><<<256 * w, 256>>>(d, 1.0f); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; r++) { (void)hipEventRecord(e0); k<T><<<256 * w, 256>>>(d, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("pattern %d waves/SIMD %d : %.2f ns per instr per SIMD\n", T, w, best * 1e6 / ((double)ITERS * 64 * w));
    (void)hipFree(d);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) { run<0>(w); run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<5>(w); run<6>(w); run<7>(w); }
    return 0;
}
