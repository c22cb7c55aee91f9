// iq_decode.h -- per-component decode of the IQ sample formats, shared by the stand-alone converter
// (io_formats.hip) and the OFDM demodulator's fused raw-input loader (ofdm_demod.hip) so that both apply
// the same operations in the same order (reference: examples/app_helpers/app_iq_readers.h:17-86,
// app_wav_reader.h:271-456).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dabgpu {

enum comp_kind { K_U8, K_S8, K_S16, K_U16, K_S24, K_S32, K_U32, K_F32, K_F64, K_ALAW, K_MULAW };

template <int NW> struct raw_words { uint32_t w[NW]; };

template <int NW>
__device__ __forceinline__ uint32_t get_byte(const raw_words<NW>& r, int i) { return (r.w[i >> 2] >> (8 * (i & 3))) & 0xFFu; }

// component k of the thread's words, as the little-endian integer the reference sees after its ReverseEndian stage
template <comp_kind K, int S, bool BE, int NW>
__device__ __forceinline__ float decode(const raw_words<NW>& r, int k) {
    if constexpr (S == 1) {
        const uint32_t v = get_byte(r, k);
        if constexpr (K == K_U8) {
            return __fmul_rn(__fsub_rn((float)v, 127.5f), 1.0f / 127.5f);
        } else if constexpr (K == K_S8) {
            return __fmul_rn((float)(int)(int8_t)v, 1.0f / 127.0f);
        } else if constexpr (K == K_ALAW) {                        // app_wav_reader.h:408-428
            const uint32_t x = v ^ 0x55u;
            const uint32_t sign = ((x >> 7) ^ 1u) & 1u, e = (x >> 4) & 7u, m = x & 15u;
            int d = (int)((m << 1) | 1u);
            if (e > 0) d |= 1 << 5;
            if (e > 1) d <<= (e - 1);
            if (sign) d = (int)(int16_t)(d ^ 0xFFFF);
            return __fmul_rn((float)d, 1.0f / 4096.0f);
        } else {                                                   // mu-law, app_wav_reader.h:436-452
            const uint32_t x = v ^ 0xFFu;
            const uint32_t sign = x >> 7, e = (x >> 4) & 7u, m = x & 15u;
            int d = (int)(((1u << 5) | (m << 1) | 1u) << e);
            if (sign) d = (int)(int16_t)(d ^ 0xFFFF);
            return __fmul_rn((float)d, 1.0f / 8192.0f);
        }
    } else if constexpr (S == 2) {
        uint32_t v = (r.w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if constexpr (BE) v = ((v & 0xFFu) << 8) | (v >> 8);
        if constexpr (K == K_S16) return __fmul_rn((float)(int)(int16_t)v, 1.0f / 32767.0f);
        else return __fmul_rn(__fsub_rn((float)v, 32767.5f), 1.0f / 32767.5f);
    } else if constexpr (S == 3) {
        uint32_t v = get_byte(r, 3 * k) | (get_byte(r, 3 * k + 1) << 8) | (get_byte(r, 3 * k + 2) << 16);
        const int s = (int)(v << 8) >> 8;
        return __fmul_rn((float)s, 1.0f / 8388607.0f);
    } else if constexpr (S == 4) {
        uint32_t v = r.w[k];
        if constexpr (BE) v = __builtin_bswap32(v);
        // float(INT32_MAX) and float(UINT32_MAX/2) + 0.5f both round to 2^31
        if constexpr (K == K_S32) return __fmul_rn((float)(int)v, 1.0f / 2147483648.0f);
        else if constexpr (K == K_U32) return __fmul_rn(__fsub_rn((float)v, 2147483648.0f), 1.0f / 2147483648.0f);
        else return __uint_as_float(v);
    } else {
        uint32_t lo = r.w[2 * k], hi = r.w[2 * k + 1];
        if constexpr (BE) { const uint32_t t = __builtin_bswap32(lo); lo = __builtin_bswap32(hi); hi = t; }
        return __double2float_rn(__hiloint2double((int)hi, (int)lo));
    }
}

}  // namespace dabgpu
