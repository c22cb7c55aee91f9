// lds_rate.hip -- development microbenchmark (round 2): cost of each LDS instruction form on gfx950 with the whole chip busy
// (16 waves per CU = 4 per SIMD), N instructions back to back per wave, conflict-free lane-linear addresses.
// Reports LDS-pipe cycles per wave-instruction per CU = wall time x clock / (instructions per CU).
// build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
enum { W_B8, W_B16, W_B32, W_B64, W_B96, W_B128, W2_B32, W2_B64, W2ST64_B64, R_B32, R_B64, R_B128, R2_B32, R2_B64, R2ST64_B64, R_U8, N_T };
static const char* names[] = {"ds_write_b8", "ds_write_b16", "ds_write_b32", "ds_write_b64", "ds_write_b96", "ds_write_b128", "ds_write2_b32", "ds_write2_b64 (adjacent)",
                              "ds_write2st64_b64", "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read2_b32", "ds_read2_b64 (adjacent)", "ds_read2st64_b64", "ds_read_u8"};
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int T> __global__ __launch_bounds__(256) void k(float* out) {
    __shared__ __attribute__((aligned(16))) char lds[256 * 32 + 2048];
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
    const unsigned t = threadIdx.x;
    f4 q = {1.f + t, 2.f, 3.f, 4.f}; f3 q3 = {1.f, 2.f, 3.f}; f2 q2 = {1.f + t, 2.f}; float q1 = t; f4 r = {0, 0, 0, 0};
    for (int it = 0; it < ITERS; it++) {
#define X8(S) S S S S S S S S
        if constexpr (T == W_B8) { const unsigned a = base + t; X8(asm volatile("ds_write_b8 %0, %1" :: "v"(a), "v"(q1) : "memory");) }
        if constexpr (T == W_B16) { const unsigned a = base + 2 * t; X8(asm volatile("ds_write_b16 %0, %1" :: "v"(a), "v"(q1) : "memory");) }
        if constexpr (T == W_B32) { const unsigned a = base + 4 * t; X8(asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(q1) : "memory");) }
        if constexpr (T == W_B64) { const unsigned a = base + 8 * t; X8(asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(q2) : "memory");) }
        if constexpr (T == W_B96) { const unsigned a = base + 16 * t; X8(asm volatile("ds_write_b96 %0, %1" :: "v"(a), "v"(q3) : "memory");) }
        if constexpr (T == W_B128) { const unsigned a = base + 16 * t; X8(asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(q) : "memory");) }
        if constexpr (T == W2_B32) { const unsigned a = base + 8 * t; X8(asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(a), "v"(q1), "v"(q2.y) : "memory");) }
        if constexpr (T == W2_B64) { const unsigned a = base + 16 * t; X8(asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(a), "v"(q2), "v"(q2) : "memory");) }
        if constexpr (T == W2ST64_B64) { const unsigned a = base + 8 * t; X8(asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:4" :: "v"(a), "v"(q2), "v"(q2) : "memory");) }
        if constexpr (T == R_B32) { const unsigned a = base + 4 * t; X8(asm volatile("ds_read_b32 %0, %1" : "=v"(r.x) : "v"(a) : "memory");) }
        if constexpr (T == R_U8) { const unsigned a = base + t; X8(asm volatile("ds_read_u8 %0, %1" : "=v"(r.x) : "v"(a) : "memory");) }
        if constexpr (T == R_B64) { const unsigned a = base + 8 * t; f2 rr; X8(asm volatile("ds_read_b64 %0, %1" : "=v"(rr) : "v"(a) : "memory");) r.x = rr.x; }
        if constexpr (T == R_B128) { const unsigned a = base + 16 * t; X8(asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a) : "memory");) }
        if constexpr (T == R2_B32) { const unsigned a = base + 8 * t; f2 rr; X8(asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(rr) : "v"(a) : "memory");) r.x = rr.x; }
        if constexpr (T == R2_B64) { const unsigned a = base + 16 * t; X8(asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(r) : "v"(a) : "memory");) }
        if constexpr (T == R2ST64_B64) { const unsigned a = base + 8 * t; X8(asm volatile("ds_read2st64_b64 %0, %1 offset1:4" : "=v"(r) : "v"(a) : "memory");) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + t] = r.x + r.y + q.x;
}
template <int T> void run() {
    float* d; (void)hipMalloc(&d, 1024 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wg_per_cu : {1, 4}) {
        const int blocks = 256 * wg_per_cu;
        k<T><<<blocks, 256>>>(d); (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; r++) { (void)hipEventRecord(e0); k<T><<<blocks, 256>>>(d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        const double instr_per_cu = (double)ITERS * 8 * 4 * wg_per_cu;
        printf("%-28s %2d waves/CU  %.3f ms  %.2f ns per wave-instr per CU (= %.1f cycles @2.1GHz)\n", names[T], 4 * wg_per_cu, best, best * 1e6 / instr_per_cu, best * 1e6 / instr_per_cu * 2.1);
    }
    (void)hipFree(d);
}
int main() {
    run<W_B8>(); run<W_B16>(); run<W_B32>(); run<W_B64>(); run<W_B96>(); run<W_B128>(); run<W2_B32>(); run<W2_B64>(); run<W2ST64_B64>();
    run<R_U8>(); run<R_B32>(); run<R_B64>(); run<R_B128>(); run<R2_B32>(); run<R2_B64>(); run<R2ST64_B64>();
    return 0;
}
