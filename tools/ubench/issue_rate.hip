// issue_rate.hip -- development microbenchmark (round 2): cycles per wave-instruction and SIMD for the exact instruction forms of
// ofdm_demod_kernel's loop body, measured with s_memtime inside the kernel (shader clock, independent of DVFS), 16 independent
// accumulators, 1 / 2 / 4 waves per SIMD.  Also: the same VALU stream with LDS stores / loads interleaved.
// build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 1024
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

enum { T_FMAC = 0, T_FMA3, T_FMAAK, T_FMA_S, T_ADD, T_MUL_NEG, T_BANK, T_RNDNE, T_CVT, T_RCP, T_CNDMASK, T_MOV, T_SWAP32, T_DPP,
       T_FMAC_W2B64, T_FMAC_W64, T_FMAC_R2B64, T_FMAC_WB8, T_FMAC_W128, T_MIX_PLL, T_COUNT };
static const char* names[] = {"v_fmac_f32_e32", "v_fma_f32 (vop3)", "v_fmaak_f32 (literal)", "v_fma_f32 sgpr c", "v_add_f32_e32", "v_mul_f32_e64 neg",
                              "v_fmac same-bank srcs", "v_rndne_f32", "v_cvt_i32_f32", "v_rcp_f32", "v_cndmask_b32", "v_mov_b32", "v_permlane32_swap",
                              "v_mov_b32_dpp quad_perm", "16 fmac + ds_write2_b64", "16 fmac + ds_write_b64", "16 fmac + ds_read2_b64",
                              "16 fmac + ds_write_b8", "16 fmac + ds_write_b128", "pll mix (fmaak/mul/add/sub)"};

template <int T>
__global__ __launch_bounds__(256) void k(float* out, uint64_t* cyc, float a, float b) {
    __shared__ float lds[256 * 8 + 64];
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
    float c = a * 3.0f, d = b;
    const unsigned la = (unsigned)(threadIdx.x * 16);      // byte address, 16-B slots
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#define X(i) "+v"(x[i])
#define ALLX X(0), X(1), X(2), X(3), X(4), X(5), X(6), X(7), X(8), X(9), X(10), X(11), X(12), X(13), X(14), X(15)
#define BODY16(S) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) : ALLX : "v"(c), "v"(d), "s"(a), "v"(la) : "memory", "vcc")
        if constexpr (T == T_FMAC) {
#define S(i) "v_fmac_f32_e32 %" #i ", %16, %17\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_FMA3) {
#define S(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_FMAAK) {
#define S(i) "v_fmaak_f32 %" #i ", %" #i ", %16, 0x3f800347\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_FMA_S) {
#define S(i) "v_fma_f32 %" #i ", %" #i ", %16, %18\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_ADD) {
#define S(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_MUL_NEG) {
#define S(i) "v_mul_f32_e64 %" #i ", %" #i ", -%16\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_BANK) {
            // explicit registers: sources v[n], v[n+4], v[n+8] share a bank if banks are index mod 4
            asm volatile(
                "v_fmac_f32_e32 v40, v44, v48\n\tv_fmac_f32_e32 v41, v45, v49\n\tv_fmac_f32_e32 v42, v46, v50\n\tv_fmac_f32_e32 v43, v47, v51\n\t"
                "v_fmac_f32_e32 v52, v44, v48\n\tv_fmac_f32_e32 v53, v45, v49\n\tv_fmac_f32_e32 v54, v46, v50\n\tv_fmac_f32_e32 v55, v47, v51\n\t"
                "v_fmac_f32_e32 v56, v44, v48\n\tv_fmac_f32_e32 v57, v45, v49\n\tv_fmac_f32_e32 v58, v46, v50\n\tv_fmac_f32_e32 v59, v47, v51\n\t"
                "v_fmac_f32_e32 v60, v44, v48\n\tv_fmac_f32_e32 v61, v45, v49\n\tv_fmac_f32_e32 v62, v46, v50\n\tv_fmac_f32_e32 v63, v47, v51\n\t"
                ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                    "v59", "v60", "v61", "v62", "v63", "memory");
            asm volatile(
                "v_fmac_f32_e32 v40, v44, v48\n\tv_fmac_f32_e32 v41, v45, v49\n\tv_fmac_f32_e32 v42, v46, v50\n\tv_fmac_f32_e32 v43, v47, v51\n\t"
                "v_fmac_f32_e32 v52, v44, v48\n\tv_fmac_f32_e32 v53, v45, v49\n\tv_fmac_f32_e32 v54, v46, v50\n\tv_fmac_f32_e32 v55, v47, v51\n\t"
                "v_fmac_f32_e32 v56, v44, v48\n\tv_fmac_f32_e32 v57, v45, v49\n\tv_fmac_f32_e32 v58, v46, v50\n\tv_fmac_f32_e32 v59, v47, v51\n\t"
                "v_fmac_f32_e32 v60, v44, v48\n\tv_fmac_f32_e32 v61, v45, v49\n\tv_fmac_f32_e32 v62, v46, v50\n\tv_fmac_f32_e32 v63, v47, v51\n\t"
                ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                    "v59", "v60", "v61", "v62", "v63", "memory");
            asm volatile(
                "v_fmac_f32_e32 v40, v44, v48\n\tv_fmac_f32_e32 v41, v45, v49\n\tv_fmac_f32_e32 v42, v46, v50\n\tv_fmac_f32_e32 v43, v47, v51\n\t"
                "v_fmac_f32_e32 v52, v44, v48\n\tv_fmac_f32_e32 v53, v45, v49\n\tv_fmac_f32_e32 v54, v46, v50\n\tv_fmac_f32_e32 v55, v47, v51\n\t"
                "v_fmac_f32_e32 v56, v44, v48\n\tv_fmac_f32_e32 v57, v45, v49\n\tv_fmac_f32_e32 v58, v46, v50\n\tv_fmac_f32_e32 v59, v47, v51\n\t"
                "v_fmac_f32_e32 v60, v44, v48\n\tv_fmac_f32_e32 v61, v45, v49\n\tv_fmac_f32_e32 v62, v46, v50\n\tv_fmac_f32_e32 v63, v47, v51\n\t"
                ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                    "v59", "v60", "v61", "v62", "v63", "memory");
            asm volatile(
                "v_fmac_f32_e32 v40, v44, v48\n\tv_fmac_f32_e32 v41, v45, v49\n\tv_fmac_f32_e32 v42, v46, v50\n\tv_fmac_f32_e32 v43, v47, v51\n\t"
                "v_fmac_f32_e32 v52, v44, v48\n\tv_fmac_f32_e32 v53, v45, v49\n\tv_fmac_f32_e32 v54, v46, v50\n\tv_fmac_f32_e32 v55, v47, v51\n\t"
                "v_fmac_f32_e32 v56, v44, v48\n\tv_fmac_f32_e32 v57, v45, v49\n\tv_fmac_f32_e32 v58, v46, v50\n\tv_fmac_f32_e32 v59, v47, v51\n\t"
                "v_fmac_f32_e32 v60, v44, v48\n\tv_fmac_f32_e32 v61, v45, v49\n\tv_fmac_f32_e32 v62, v46, v50\n\tv_fmac_f32_e32 v63, v47, v51\n\t"
                ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                    "v59", "v60", "v61", "v62", "v63", "memory");
        } else if constexpr (T == T_RNDNE) {
#define S(i) "v_rndne_f32_e32 %" #i ", %" #i "\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_CVT) {
#define S(i) "v_cvt_i32_f32_e32 %" #i ", %" #i "\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_RCP) {
#define S(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_CNDMASK) {
#define S(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %16, vcc\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_MOV) {
#define S(i) "v_mov_b32_e32 %" #i ", %16\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_SWAP32) {
#define S(i) "v_permlane32_swap_b32_e32 %" #i ", %16\n\t"
            asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) : ALLX, "+v"(c) : "v"(c), "v"(d), "s"(a), "v"(la) : "memory");
            asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) : ALLX, "+v"(c) : "v"(c), "v"(d), "s"(a), "v"(la) : "memory");
            asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) : ALLX, "+v"(c) : "v"(c), "v"(d), "s"(a), "v"(la) : "memory");
            asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) : ALLX, "+v"(c) : "v"(c), "v"(d), "s"(a), "v"(la) : "memory");
#undef S
        } else if constexpr (T == T_DPP) {
#define S(i) "v_mov_b32_dpp %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            BODY16(S); BODY16(S); BODY16(S); BODY16(S);
#undef S
        } else if constexpr (T == T_FMAC_W2B64 || T == T_FMAC_W64 || T == T_FMAC_R2B64 || T == T_FMAC_WB8 || T == T_FMAC_W128) {
#define S(i) "v_fmac_f32_e32 %" #i ", %16, %17\n\t"
            for (int r = 0; r < 4; r++) {
                BODY16(S);
                if constexpr (T == T_FMAC_W2B64) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(la), "v"(*(double*)&x[0]), "v"(*(double*)&x[2]) : "memory");
                if constexpr (T == T_FMAC_W64) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(*(double*)&x[0]) : "memory");
                if constexpr (T == T_FMAC_WB8) asm volatile("ds_write_b8 %0, %1" :: "v"(la), "v"(x[0]) : "memory");
                if constexpr (T == T_FMAC_W128) { typedef float f4 __attribute__((ext_vector_type(4))); f4 q = {x[0], x[1], x[2], x[3]};
                    asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(q) : "memory"); }
                if constexpr (T == T_FMAC_R2B64) { typedef float f4 __attribute__((ext_vector_type(4))); f4 q;
                    asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(2)" : "=v"(q) : "v"(la) : "memory"); d += q.x * 0.0f; }
            }
#undef S
        } else if constexpr (T == T_MIX_PLL) {
            // the shape of cheb2: z = x*x; 5 x fmaak; sub; mul; mul  (9 ops per value, 16 values)
#define S(i) "v_mul_f32_e32 %" #i ", %" #i ", %" #i "\n\t"
            BODY16(S);
#undef S
#define S(i) "v_fmaak_f32 %" #i ", %" #i ", %16, 0x3f800347\n\t"
            BODY16(S); BODY16(S);
#undef S
#define S(i) "v_sub_f32_e32 %" #i ", %" #i ", %17\n\t"
            BODY16(S);
#undef S
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = c + d;
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T> void run(int waves_per_simd) {
    float* d; uint64_t* c;
    const int blocks = 256 * waves_per_simd;       // 256-thread blocks: 1 wave per SIMD each
    hipMalloc(&d, blocks * 256 * sizeof(float)); hipMalloc(&c, blocks * 4 * sizeof(uint64_t));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T><<<blocks, 256>>>(d, c, 1.0001f, 1e-7f); hipDeviceSynchronize();
    hipEventRecord(e0); k<T><<<blocks, 256>>>(d, c, 1.0001f, 1e-7f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t* h = (uint64_t*)malloc(blocks * 4 * sizeof(uint64_t));
    hipMemcpy(h, c, blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks * 4; i++) sum += (double)h[i];
    const double per_wave = sum / (blocks * 4);
    const double n_valu = (double)ITERS * 64;
    // s_memtime counts at a constant 100 MHz on this part; convert through the wall time of the launch instead
    const double insts_per_simd = n_valu * waves_per_simd;
    printf("%-30s w/SIMD=%d  %.3f ms  %.2f ns per VALU wave-instr per SIMD  (memtime ticks per wave %.0f)\n", names[T], waves_per_simd, ms,
           ms * 1e6 / insts_per_simd, per_wave);
    free(h); hipFree(d); hipFree(c);
}
template <int T> void runall() { for (int w : {1, 2, 4}) run<T>(w); }
int main() {
    runall<T_FMAC>(); runall<T_FMA3>(); runall<T_FMAAK>(); runall<T_FMA_S>(); runall<T_ADD>(); runall<T_MUL_NEG>(); runall<T_BANK>();
    runall<T_RNDNE>(); runall<T_CVT>(); runall<T_RCP>(); runall<T_CNDMASK>(); runall<T_MOV>(); runall<T_SWAP32>(); runall<T_DPP>();
    runall<T_FMAC_W2B64>(); runall<T_FMAC_W64>(); runall<T_FMAC_R2B64>(); runall<T_FMAC_WB8>(); runall<T_FMAC_W128>(); runall<T_MIX_PLL>();
    return 0;
}
