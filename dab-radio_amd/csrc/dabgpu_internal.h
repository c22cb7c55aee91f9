// dabgpu_internal.h -- shared between the translation units of libdabgpu.so (not installed)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <vector>

struct dabgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<float> prs;          // host copy, 2*2048
    std::vector<int> mapper;         // host copy, 1536
    float* d_tw = nullptr;           // 2048 x (cos, -sin)
    uint16_t* d_inv_map = nullptr;   // carrier -> de-interleaved position
    float* d_prs = nullptr;          // PRS spectrum
    std::vector<void*> scratch;      // grow-only device scratch slots
    std::vector<size_t> scratch_bytes;
};

void dabgpu_set_error(const char* fmt, ...);
int dabgpu_check_hip(hipError_t e, const char* what);
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out);

extern "C" hipError_t dabgpu_launch_ofdm_demod(const float* d_iq, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                               float* d_fft, const float* d_tw, const uint16_t* d_inv_map,
                                               int n_frames, int sym_per_chunk, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_ofdm_phase(const float* d_cp_corr, int n_frames, float beta, float* d_total_phase,
                                               float* d_fine_freq, hipStream_t stream);
