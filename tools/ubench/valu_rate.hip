// valu_rate.hip -- development microbenchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ACC 16
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x[N_ACC]; f2 y[N_ACC];
    for (int i = 0; i < N_ACC; i++) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
            if (MODE == 2) x[i] = x[i] * a;
            if (MODE == 3) y[i] = y[i] * a2;
            if (MODE == 4) x[i] = x[i] + a;
            if (MODE == 5) y[i] = y[i] + a2;
            if (MODE == 6) { x[i] = __builtin_fmaf(x[i], a, b); y[i] = __builtin_elementwise_fma(y[i], a2, b2); }
        }
    }
    float s = 0; for (int i = 0; i < N_ACC; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int ops_per_iter_per_lane, int waves_per_simd) {
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float) * 4);
    const int blocks = 256 * waves_per_simd;       // 256-thread blocks: 1 wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 1e-7f); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 1.0001f, 1e-7f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 /*waves*/ * ITERS * N_ACC * (MODE == 6 ? 2 : 1);
    const double per_simd = insts / 1024.0;   // wave-instructions per SIMD
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz), %.1f TFLOP-equivalent/s\n",
           name, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
           insts * 64 * ops_per_iter_per_lane / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", 2, w); run<1>("v_pk_fma_f32", 4, w); run<2>("v_mul_f32", 1, w); run<3>("v_pk_mul_f32", 2, w);
        run<4>("v_add_f32", 1, w); run<5>("v_pk_add_f32", 2, w); run<6>("v_fma + v_pk_fma interleaved", 3, w);
    }
    return 0;
}
