// lds_dma_probe.hip -- development probe: where does global_load_lds_dwordx4 (gfx950) put each lane's 16 bytes?
// build: hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned* __restrict__ g, unsigned* out) {
    __shared__ unsigned s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0xDEADBEEFu;
    __syncthreads();
    // every lane supplies its own global address (lane * 16 bytes, reversed to see which lane lands where)
    const unsigned* src = g + 4 * (63 - threadIdx.x);
    __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)&s[32], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
int main() {
    unsigned h[256], *d, *o, r[1024];
    for (int i = 0; i < 256; i++) h[i] = i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 24; i < 48; i++) printf("s[%d]=%x ", i, r[i]);
    printf("\n... ");
    for (int i = 280; i < 296; i++) printf("s[%d]=%x ", i, r[i]);
    printf("\n");
    int first = -1, last = -1;
    for (int i = 0; i < 1024; i++) if (r[i] != 0xDEADBEEFu) { if (first < 0) first = i; last = i; }
    printf("written range [%d, %d]\n", first, last);
    return 0;
}
