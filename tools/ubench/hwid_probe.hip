// hwid_probe.hip -- development probe: where do the single-wavefront workgroups of a launch land?  Every workgroup records
// HW_ID (wave slot, SIMD, CU, SE, XCC) while all of them are resident; the host prints the distribution of wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
// HOG: make the wavefront as big as the lane-mapped Viterbi (128 VGPRs, 4 wavefronts per SIMD at most)
template <int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_hog(unsigned* out, int spin, unsigned a) {
    unsigned x[100];
#pragma unroll
    for (int i = 0; i < 100; i++) x[i] = threadIdx.x * 7u + i + a;
    unsigned id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    for (int k = 0; k < spin; k++) {
        __builtin_amdgcn_s_sleep(127);
#pragma unroll
        for (int i = 0; i < 100; i++) x[i] = x[i] * 1664525u + x[(i + 1) % 100];
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 100; i++) s += x[i];
    const int w = blockIdx.x * WPB + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) { out[2 * w] = id; out[2 * w + 1] = xcc | (s == 0x1234567u ? 16u : 0u); }
}
__global__ __launch_bounds__(64) void probe(unsigned* out, int spin, int vgpr_hog) {
    unsigned id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);          // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);          // HW_REG_XCC_ID
    for (int k = 0; k < spin; k++) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = id; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    const int counts[] = {288, 576, 1024, 1152, 288, 576, 1024, 1152, 288, 576, 1024, 1152};
    int idx = 0;
    for (int n : counts) {
        unsigned* d; hipMalloc(&d, n * 2 * sizeof(unsigned));
        const int variant = idx++ / 4;
        if (variant == 0) probe<<<n, 64>>>(d, 30, 0);
        else if (variant == 1) probe_hog<1><<<n, 64>>>(d, 30, 3u);
        else probe_hog<4><<<n / 4, 256>>>(d, 30, 3u);
        printf(variant == 0 ? "small wave   " : variant == 1 ? "128-VGPR wave" : "128-VGPR, 4 waves per workgroup");
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_simd;
        for (int i = 0; i < n; i++) {
            const unsigned id = h[2 * i], xcc = h[2 * i + 1] & 15u;
            const unsigned simd = (id >> 4) & 3u, cu = (id >> 8) & 15u, sh = (id >> 12) & 1u, se = (id >> 13) & 7u;
            per_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
        }
        std::map<int, int> hist;
        for (auto& kv : per_simd) hist[kv.second]++;
        printf("%5d workgroups: %zu distinct SIMDs used; waves-per-SIMD histogram:", n, per_simd.size());
        for (auto& kv : hist) printf("  %d waves x %d SIMDs", kv.first, kv.second);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
