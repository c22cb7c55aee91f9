// load_scaling.hip -- development probe: does the time of ONE wavefront per SIMD depend on how many SIMDs of the chip are busy?
// A register-only packed-integer loop (the instruction mix of the lane-mapped Viterbi: v_pk_add_u16 / v_pk_min_i16 / v_pk_sub_i16 /
// v_perm_b32) in single-wavefront workgroups; grids of 128 ... 4096 workgroups.  Clock throttling shows up as time growing with the
// grid below 1024 workgroups (1 per SIMD); 2048 / 4096 = 2 / 4 per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 load_scaling.hip -o load_scaling
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef UNR
#define UNR 40
#endif
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void body(unsigned* out, unsigned a, int iters, unsigned* sink, int with_stores) {
    unsigned x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 77u + i + blockIdx.x;
    for (int it = 0; it < iters; it += UNR) {
_Pragma("unroll") for (int uu = 0; uu < UNR; uu++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            s2 v = __builtin_bit_cast(s2, x[i]), w = __builtin_bit_cast(s2, x[(i + 5) & 15] ^ a);
            us2 s = __builtin_bit_cast(us2, v) + __builtin_bit_cast(us2, w);
            s2 m = __builtin_elementwise_min(__builtin_bit_cast(s2, s), w);
            s2 d = __builtin_elementwise_sub_sat(m, v);
            x[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, m), 0x07050301u);
        }
        if (with_stores) {
            uint4 q; q.x = x[0]; q.y = x[1]; q.z = x[2]; q.w = x[3];
            reinterpret_cast<uint4*>(sink + ((size_t)blockIdx.x * 4096 + (size_t)((it + uu) & 4095)) * 256)[threadIdx.x] = q;
        }
    }
    }
    unsigned s = 0; for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    unsigned *d, *sink; hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&sink, (size_t)4096 * 4096 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ws = 0; ws < 1; ws++) for (int grid : {128, 288, 512, 1024, 2048, 4096}) {
        body<<<grid, 64>>>(d, 0x00030005u, 2000, sink, ws); hipDeviceSynchronize();
        hipEventRecord(e0); body<<<grid, 64>>>(d, 0x00030005u, 20000, sink, ws); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stores=%d grid %5d: %.3f ms  (%.2f ns per 64-instruction iteration)\n", ws, grid, ms, ms * 1e6 / 20000);
    }
    return 0;
}
