// io_formats.hip -- the data formats either side of the hot path, on the device:
//   * IQ sample formats -> interleaved complex float, with the arithmetic of the reference's readers
//     (examples/app_helpers/app_iq_readers.h:17-159, app_wav_reader.h:257-470);
//   * soft-bit <-> packed hard-bit conversion (examples/app_helpers/app_viterbi_convert_block.h:12-44);
//   * the host-side wav header walk (app_wav_reader.h:107-255) that tells the converter where the samples start.
// All kernels are pure streaming (HBM bound): one thread produces one 16-byte store, loads are 4..32 bytes per lane,
// both fully coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "dabgpu.h"
#include "dabgpu_internal.h"
#include "iq_decode.h"

using namespace dabgpu;

namespace {

// ---- component descriptions -------------------------------------------------------------------------------
struct fmt_info { comp_kind kind; int size; bool big_endian; };

constexpr fmt_info FMT_TABLE[DABGPU_IQ_NB_FORMATS] = {
    {K_U8, 1, false}, {K_S8, 1, false},
    {K_S16, 2, false}, {K_S16, 2, true}, {K_U16, 2, false}, {K_U16, 2, true},
    {K_S32, 4, false}, {K_S32, 4, true}, {K_U32, 4, false}, {K_U32, 4, true},
    {K_F32, 4, false}, {K_F32, 4, true}, {K_F64, 8, false}, {K_F64, 8, true},
    {K_U8, 1, false},                 // WAV_PCM8:  (v - 255/2.0f) * (1/(255/2.0f)): the raw_u8 arithmetic
    {K_S16, 2, false},                // WAV_PCM16: v * (1/32767.f): the raw_s16l arithmetic
    {K_S24, 3, false},
    {K_S32, 4, false},                // WAV_PCM32: v * (1/float(INT32_MAX)): the raw_s32l arithmetic
    {K_F32, 4, false}, {K_F64, 8, false}, {K_ALAW, 1, false}, {K_MULAW, 1, false},
};

constexpr int COMPS_PER_THREAD = 4;   // 2 IQ samples -> one float4 store

template <comp_kind K, int S, bool BE>
__global__ __launch_bounds__(256) void iq_convert_kernel(const uint8_t* __restrict__ raw, float* __restrict__ out, size_t n_comp) {
    constexpr int NB = COMPS_PER_THREAD * S;          // bytes per thread: 4, 8, 12, 16 or 32
    constexpr int NW = NB / 4;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t c0 = t * COMPS_PER_THREAD;
    if (c0 >= n_comp) return;
    raw_words<NW> r;
    const uint8_t* p = raw + c0 * S;
    if (c0 + COMPS_PER_THREAD <= n_comp) {
        if constexpr (NW == 1) r.w[0] = *reinterpret_cast<const uint32_t*>(p);
        else if constexpr (NW == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); r.w[0] = v.x; r.w[1] = v.y; }
        else if constexpr (NW == 3) {                 // 12-byte items are only 4-byte aligned
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
            r.w[0] = q[0]; r.w[1] = q[1]; r.w[2] = q[2];
        } else {
#pragma unroll
            for (int i = 0; i < NW / 4; i++) {
                const uint4 v = reinterpret_cast<const uint4*>(p)[i];
                r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
            }
        }
        float4 o;
        o.x = decode<K, S, BE>(r, 0); o.y = decode<K, S, BE>(r, 1);
        o.z = decode<K, S, BE>(r, 2); o.w = decode<K, S, BE>(r, 3);
        *reinterpret_cast<float4*>(out + c0) = o;
    } else {                                          // ragged tail: 2 components (one IQ sample)
        const int n_left = (int)(n_comp - c0);
#pragma unroll
        for (int i = 0; i < NW; i++) r.w[i] = 0;
        for (int i = 0; i < n_left * S; i++) r.w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
        const float v0 = decode<K, S, BE>(r, 0), v1 = decode<K, S, BE>(r, 1), v2 = decode<K, S, BE>(r, 2);
        if (n_left > 0) out[c0] = v0;
        if (n_left > 1) out[c0 + 1] = v1;
        if (n_left > 2) out[c0 + 2] = v2;
    }
}

template <comp_kind K, int S, bool BE>
hipError_t launch_convert(const void* d_raw, float* d_iq, size_t n_comp, hipStream_t s) {
    const size_t n_threads = (n_comp + COMPS_PER_THREAD - 1) / COMPS_PER_THREAD;
    const size_t n_blocks = (n_threads + 255) / 256;
    if (n_blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((iq_convert_kernel<K, S, BE>), dim3((unsigned)n_blocks), dim3(256), 0, s,
                       static_cast<const uint8_t*>(d_raw), d_iq, n_comp);
    return hipGetLastError();
}

// ---- soft bits <-> packed hard bits ---------------------------------------------------------------------------
// word bit k = (soft[32 w + k] >= 0): byte j of the little-endian word holds bits 8j..8j+7 LSB first, which is
// convert_viterbi_bits_to_bytes' layout (app_viterbi_convert_block.h:28-44, MID_POINT = 0)
__device__ __forceinline__ uint32_t sign_mask4(uint32_t w) {     // 4 soft bytes -> 4 bits, bit i = (byte i >= 0)
    const uint32_t nonneg = ~w & 0x80808080u;
    return ((nonneg >> 7) | (nonneg >> 14) | (nonneg >> 21) | (nonneg >> 28)) & 0xFu;
}

__global__ __launch_bounds__(256) void soft_to_hard_kernel(const int8_t* __restrict__ bits, uint8_t* __restrict__ bytes, size_t n_bytes) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b0 = t * 4;                           // 4 output bytes = 32 soft bits per thread
    if (b0 >= n_bytes) return;
    if (b0 + 4 <= n_bytes) {
        const uint4* p = reinterpret_cast<const uint4*>(bits + b0 * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w = sign_mask4(lo.x) | (sign_mask4(lo.y) << 4) | (sign_mask4(lo.z) << 8) | (sign_mask4(lo.w) << 12) |
                           (sign_mask4(hi.x) << 16) | (sign_mask4(hi.y) << 20) | (sign_mask4(hi.z) << 24) | (sign_mask4(hi.w) << 28);
        *reinterpret_cast<uint32_t*>(bytes + b0) = w;
    } else {
        for (size_t b = b0; b < n_bytes; b++) {
            uint32_t v = 0;
            for (int i = 0; i < 8; i++) v |= (uint32_t)(bits[b * 8 + i] >= 0) << i;
            bytes[b] = (uint8_t)v;
        }
    }
}

// bit i of a byte (LSB first) -> +127 / -127 (convert_viterbi_bytes_to_bits, app_viterbi_convert_block.h:12-26)
__device__ __forceinline__ uint32_t expand4(uint32_t nib) {      // 4 bits -> 4 soft bytes
    const uint32_t spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);   // 0/1 per byte
    return 0x81818181u ^ (spread * 0xFEu);                       // 0x81 = -127, 0x81 ^ 0xFE = 0x7F = +127
}

__global__ __launch_bounds__(256) void hard_to_soft_kernel(const uint8_t* __restrict__ bytes, int8_t* __restrict__ bits, size_t n_bytes) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b0 = t * 2;                           // 2 input bytes -> 16 soft bits (one 16-byte store)
    if (b0 >= n_bytes) return;
    const uint32_t v0 = bytes[b0];
    if (b0 + 1 < n_bytes) {
        const uint32_t v1 = bytes[b0 + 1];
        *reinterpret_cast<uint4*>(bits + b0 * 8) = make_uint4(expand4(v0 & 15u), expand4(v0 >> 4), expand4(v1 & 15u), expand4(v1 >> 4));
    } else {
        *reinterpret_cast<uint2*>(bits + b0 * 8) = make_uint2(expand4(v0 & 15u), expand4(v0 >> 4));
    }
}

}  // namespace

extern "C" {

int dabgpu_iq_convert(dabgpu_ctx* c, const void* d_raw, int format, size_t n_samples, float* d_iq, void* stream) {
    if (!c) { dabgpu_set_error("iq_convert: null context"); return DABGPU_ERR_INVALID_ARG; }
    if (format < 0 || format >= DABGPU_IQ_NB_FORMATS) { dabgpu_set_error("iq_convert: unknown format %d", format); return DABGPU_ERR_INVALID_ARG; }
    if (n_samples == 0) return DABGPU_OK;
    if (!d_raw || !d_iq) { dabgpu_set_error("iq_convert: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    if (((uintptr_t)d_raw | (uintptr_t)d_iq) & 15u) { dabgpu_set_error("iq_convert: buffers must be 16-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = 2 * n_samples;
    const fmt_info f = FMT_TABLE[format];
    hipError_t e = hipErrorInvalidValue;
    switch (f.kind) {
    case K_U8: e = launch_convert<K_U8, 1, false>(d_raw, d_iq, n, s); break;
    case K_S8: e = launch_convert<K_S8, 1, false>(d_raw, d_iq, n, s); break;
    case K_ALAW: e = launch_convert<K_ALAW, 1, false>(d_raw, d_iq, n, s); break;
    case K_MULAW: e = launch_convert<K_MULAW, 1, false>(d_raw, d_iq, n, s); break;
    case K_S16: e = f.big_endian ? launch_convert<K_S16, 2, true>(d_raw, d_iq, n, s) : launch_convert<K_S16, 2, false>(d_raw, d_iq, n, s); break;
    case K_U16: e = f.big_endian ? launch_convert<K_U16, 2, true>(d_raw, d_iq, n, s) : launch_convert<K_U16, 2, false>(d_raw, d_iq, n, s); break;
    case K_S24: e = launch_convert<K_S24, 3, false>(d_raw, d_iq, n, s); break;
    case K_S32: e = f.big_endian ? launch_convert<K_S32, 4, true>(d_raw, d_iq, n, s) : launch_convert<K_S32, 4, false>(d_raw, d_iq, n, s); break;
    case K_U32: e = f.big_endian ? launch_convert<K_U32, 4, true>(d_raw, d_iq, n, s) : launch_convert<K_U32, 4, false>(d_raw, d_iq, n, s); break;
    case K_F32: e = f.big_endian ? launch_convert<K_F32, 4, true>(d_raw, d_iq, n, s) : launch_convert<K_F32, 4, false>(d_raw, d_iq, n, s); break;
    case K_F64: e = f.big_endian ? launch_convert<K_F64, 8, true>(d_raw, d_iq, n, s) : launch_convert<K_F64, 8, false>(d_raw, d_iq, n, s); break;
    }
    return dabgpu_check_hip(e, "iq_convert_kernel launch");
}

int dabgpu_soft_bits_to_hard_bytes(dabgpu_ctx* c, const int8_t* d_bits, size_t n_bytes, uint8_t* d_bytes, void* stream) {
    if (!c) { dabgpu_set_error("soft_bits_to_hard_bytes: null context"); return DABGPU_ERR_INVALID_ARG; }
    if (n_bytes == 0) return DABGPU_OK;
    if (!d_bits || !d_bytes) { dabgpu_set_error("soft_bits_to_hard_bytes: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    if (((uintptr_t)d_bits & 15u) || ((uintptr_t)d_bytes & 3u)) { dabgpu_set_error("soft_bits_to_hard_bytes: misaligned buffer"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    const size_t n_blocks = ((n_bytes + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(soft_to_hard_kernel, dim3((unsigned)n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_bits, d_bytes, n_bytes);
    return dabgpu_check_hip(hipGetLastError(), "soft_to_hard_kernel launch");
}

int dabgpu_hard_bytes_to_soft_bits(dabgpu_ctx* c, const uint8_t* d_bytes, size_t n_bytes, int8_t* d_bits, void* stream) {
    if (!c) { dabgpu_set_error("hard_bytes_to_soft_bits: null context"); return DABGPU_ERR_INVALID_ARG; }
    if (n_bytes == 0) return DABGPU_OK;
    if (!d_bits || !d_bytes) { dabgpu_set_error("hard_bytes_to_soft_bits: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    if ((uintptr_t)d_bits & 15u) { dabgpu_set_error("hard_bytes_to_soft_bits: misaligned buffer"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    const size_t n_blocks = ((n_bytes + 1) / 2 + 255) / 256;
    hipLaunchKernelGGL(hard_to_soft_kernel, dim3((unsigned)n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_bytes, d_bits, n_bytes);
    return dabgpu_check_hip(hipGetLastError(), "hard_to_soft_kernel launch");
}

// ---- host-buffer forms ------------------------------------------------------------------------------------------
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
static int round_trip(dabgpu_ctx* c, const void* h_in, size_t in_bytes, void* h_out, size_t out_bytes,
                      int (*run)(dabgpu_ctx*, const void*, void*, size_t, int, hipStream_t), size_t n, int arg) {
    int st;
    DABGPU_BIND(c);
    void *d_in, *d_out;
    if ((st = dabgpu_scratch(c, 20, in_bytes + 16, &d_in))) return st;
    if ((st = dabgpu_scratch(c, 21, out_bytes + 16, &d_out))) return st;
    hipStream_t s = c->stream;
    CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s));
    if ((st = run(c, d_in, d_out, n, arg, s))) return st;
    CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return DABGPU_OK;
}
#undef CK

int dabgpu_iq_convert_host_sync(dabgpu_ctx* c, const void* h_raw, int format, size_t n_samples, float* h_iq) {
    if (!c) { dabgpu_set_error("iq_convert_host_sync: null context"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_HOST_LOCK(c);
    const size_t sb = dabgpu_iq_format_sample_bytes(format);
    if (sb == 0) { dabgpu_set_error("iq_convert_host_sync: unknown format %d", format); return DABGPU_ERR_INVALID_ARG; }
    if (n_samples == 0) return DABGPU_OK;
    if (!h_raw || !h_iq) { dabgpu_set_error("iq_convert_host_sync: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    return round_trip(c, h_raw, n_samples * sb, h_iq, n_samples * 2 * sizeof(float),
                      [](dabgpu_ctx* c, const void* i, void* o, size_t n, int f, hipStream_t s) {
                          return dabgpu_iq_convert(c, i, f, n, static_cast<float*>(o), s); }, n_samples, format);
}

int dabgpu_soft_bits_to_hard_bytes_host_sync(dabgpu_ctx* c, const int8_t* h_bits, size_t n_bytes, uint8_t* h_bytes) {
    if (!c) { dabgpu_set_error("soft_bits_to_hard_bytes_host_sync: null context"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_HOST_LOCK(c);
    if (n_bytes == 0) return DABGPU_OK;
    if (!h_bits || !h_bytes) { dabgpu_set_error("soft_bits_to_hard_bytes_host_sync: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    return round_trip(c, h_bits, n_bytes * 8, h_bytes, n_bytes,
                      [](dabgpu_ctx* c, const void* i, void* o, size_t n, int, hipStream_t s) {
                          return dabgpu_soft_bits_to_hard_bytes(c, static_cast<const int8_t*>(i), n, static_cast<uint8_t*>(o), s); }, n_bytes, 0);
}

int dabgpu_hard_bytes_to_soft_bits_host_sync(dabgpu_ctx* c, const uint8_t* h_bytes, size_t n_bytes, int8_t* h_bits) {
    if (!c) { dabgpu_set_error("hard_bytes_to_soft_bits_host_sync: null context"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_HOST_LOCK(c);
    if (n_bytes == 0) return DABGPU_OK;
    if (!h_bits || !h_bytes) { dabgpu_set_error("hard_bytes_to_soft_bits_host_sync: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    return round_trip(c, h_bytes, n_bytes, h_bits, n_bytes * 8,
                      [](dabgpu_ctx* c, const void* i, void* o, size_t n, int, hipStream_t s) {
                          return dabgpu_hard_bytes_to_soft_bits(c, static_cast<const uint8_t*>(i), n, static_cast<int8_t*>(o), s); }, n_bytes, 0);
}

}  // extern "C"
