// iq_sample.h -- sample i of a block in the capture formats the demodulation kernels read by themselves:
// SRC 0 = complex float, 1 = raw_u8, 2 = raw_s8, 3 = raw_s16l; dequantised with the reader arithmetic of iq_decode.h
// (examples/app_helpers/app_iq_readers.h:19-44,79-84)
#pragma once
#include "iq_decode.h"
#include "ofdm_device.h"

namespace dabgpu {

template <int SRC> struct src_sample_bytes { static constexpr int value = (SRC == 0) ? 8 : (SRC == 3) ? 4 : 2; };
template <int SRC>
__device__ __forceinline__ f2 sample_at(const uint8_t* __restrict__ base, long long i) {
    const uint8_t* p = base + i * src_sample_bytes<SRC>::value;
    if constexpr (SRC == 0) {
        return *reinterpret_cast<const f2*>(p);
    } else if constexpr (SRC == 3) {
        typedef uint32_t u32a2 __attribute__((aligned(2)));
        raw_words<1> r; r.w[0] = *reinterpret_cast<const u32a2*>(p);
        return mk2(decode<K_S16, 2, false>(r, 0), decode<K_S16, 2, false>(r, 1));
    } else {
        constexpr comp_kind K = (SRC == 1) ? K_U8 : K_S8;
        typedef uint16_t u16a1 __attribute__((aligned(1)));
        raw_words<1> r; r.w[0] = *reinterpret_cast<const u16a1*>(p);
        return mk2(decode<K, 1, false>(r, 0), decode<K, 1, false>(r, 1));
    }
}

}  // namespace dabgpu
