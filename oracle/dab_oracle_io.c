/* dab_oracle_io.c -- CPU oracle (TEST INFRASTRUCTURE, never shipped, never on the product path) for the data formats
 * either side of the hot path: the reference's IQ readers and soft/hard bit converters restated sample by sample.
 * Pinned against the reference's own headers compiled in place (oracle/ref_harness_io.cpp -> oracle/_ref) by
 * tests/test_oracle_pins.py; the golden vectors under tests/golden/ carry the same comparison to the GPU box.
 */
#include <limits.h>
#include <string.h>

#include "dab_oracle.h"

enum kind { U8, S8, S16, U16, S24, S32, U32, F32, F64, ALAW, MULAW };
static const struct { enum kind k; int size; int be; } FORMATS[DAB_IQ_NB_FORMATS] = {
    {U8, 1, 0}, {S8, 1, 0}, {S16, 2, 0}, {S16, 2, 1}, {U16, 2, 0}, {U16, 2, 1}, {S32, 4, 0}, {S32, 4, 1}, {U32, 4, 0}, {U32, 4, 1},
    {F32, 4, 0}, {F32, 4, 1}, {F64, 8, 0}, {F64, 8, 1},
    {U8, 1, 0} /* set apart below: the wav PCM8 constants are spelled differently */, {S16, 2, 0}, {S24, 3, 0}, {S32, 4, 0},
    {F32, 4, 0}, {F64, 8, 0}, {ALAW, 1, 0}, {MULAW, 1, 0},
};

size_t dab_iq_component_bytes(int format) {
    return (format < 0 || format >= DAB_IQ_NB_FORMATS) ? 0 : (size_t)FORMATS[format].size;
}

/* the component's bytes in little-endian order (what reverse_endian_inplace leaves on a little-endian host,
 * app_io_buffers.h:268-284) */
static void load_le(const uint8_t *p, int size, int be, uint8_t *b) {
    for (int i = 0; i < size; i++) b[i] = be ? p[size - 1 - i] : p[i];
}

int dab_iq_convert(const uint8_t *raw, int format, size_t n_comp, float *out) {
    if (format < 0 || format >= DAB_IQ_NB_FORMATS) return -1;
    const enum kind k = FORMATS[format].k;
    const int size = FORMATS[format].size, be = FORMATS[format].be;
    const int wav = format >= 14;
    for (size_t i = 0; i < n_comp; i++) {
        uint8_t b[8];
        load_le(raw + i * (size_t)size, size, be, b);
        float y = 0.0f;
        switch (k) {
        case U8:
            if (wav) {                                   /* app_wav_reader.h:274-279 */
                const float BIAS = (float)UINT8_MAX / 2.0f;
                const float SCALE = 1.0f / BIAS;
                y = ((float)b[0] - BIAS) * SCALE;
            } else {                                     /* app_iq_readers.h:23-31,79-84 */
                const float BIAS = (float)(UINT8_MAX / 2) + 0.5f;
                const float scale = 1.0f / BIAS;
                const float v = (float)b[0] - BIAS;
                y = v * scale;
            }
            break;
        case S8: {
            const float scale = 1.0f / (float)INT8_MAX;
            y = (float)(int8_t)b[0] * scale;
        } break;
        case S16: {                                      /* raw_s16*: 1/float(int16 max); wav pcm16 :291-296 the same */
            const int16_t v = (int16_t)((uint16_t)b[0] | ((uint16_t)b[1] << 8));
            const float scale = 1.0f / (float)INT16_MAX;
            y = (float)v * scale;
        } break;
        case U16: {
            const uint16_t v = (uint16_t)(b[0] | (b[1] << 8));
            const float BIAS = (float)(UINT16_MAX / 2) + 0.5f;
            const float scale = 1.0f / BIAS;
            y = ((float)v - BIAS) * scale;
        } break;
        case S24: {                                      /* app_wav_reader.h:305-313 */
            int32_t v = (int32_t)b[0] | ((int32_t)b[1] << 8) | ((int32_t)b[2] << 16);
            if (v & 0x800000) v |= (int32_t)0xFF000000u;
            const float SCALE = 1.0f / (float)(int32_t)0x7FFFFF;
            y = (float)v * SCALE;
        } break;
        case S32: {
            const int32_t v = (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
            const float scale = 1.0f / (float)INT32_MAX;
            y = (float)v * scale;
        } break;
        case U32: {
            const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
            const float BIAS = (float)(UINT32_MAX / 2u) + 0.5f;
            const float scale = 1.0f / BIAS;
            y = ((float)v - BIAS) * scale;
        } break;
        case F32:
            memcpy(&out[i], b, 4);                       /* a reinterpret cast: the bits travel untouched */
            continue;
        case F64: {
            double d;
            memcpy(&d, b, 8);
            y = (float)d;
        } break;
        case ALAW: {                                     /* app_wav_reader.h:408-428 */
            uint8_t value = b[0];
            value ^= 0x55;
            const int16_t sign = (int16_t)((value >> 7) ^ 1);
            const uint8_t exponent = (value >> 4) & 7;
            const int16_t mantissa = (int16_t)(value & 15);
            int16_t decoded = (int16_t)((mantissa << 1) | 1);
            if (exponent > 0) decoded |= 1 << 5;
            if (exponent > 1) decoded = (int16_t)(decoded << (exponent - 1));
            if (sign) decoded = (int16_t)(decoded ^ 0xFFFF);
            y = (float)decoded * (1.0f / (float)(int16_t)0x1000);
        } break;
        case MULAW: {                                    /* app_wav_reader.h:436-452 */
            uint8_t value = b[0];
            value ^= 0xFF;
            const int16_t sign = (int16_t)(value >> 7);
            const uint8_t exponent = (value >> 4) & 7;
            const int16_t mantissa = (int16_t)(value & 15);
            int16_t decoded = (int16_t)(((1 << 5) | (mantissa << 1) | 1) << exponent);
            if (sign) decoded = (int16_t)(decoded ^ 0xFFFF);
            y = (float)decoded * (1.0f / (float)(int16_t)0x2000);
        } break;
        }
        out[i] = y;
    }
    return 0;
}

void dab_hard_bytes_to_soft_bits(const uint8_t *bytes, size_t n_bytes, int8_t *bits) {
    for (size_t j = 0; j < n_bytes; j++)
        for (int i = 0; i < 8; i++) bits[8 * j + i] = ((bytes[j] >> i) & 1) ? +127 : -127;
}

void dab_soft_bits_to_hard_bytes(const int8_t *bits, size_t n_bytes, uint8_t *bytes) {
    const int8_t MID_POINT = (+127 + -127) / 2;
    for (size_t j = 0; j < n_bytes; j++) {
        uint8_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint8_t)((bits[8 * j + i] >= MID_POINT) ? 1 : 0) << i;
        bytes[j] = v;
    }
}

static uint32_t rd(const uint8_t *p, int n) {
    uint32_t x = 0;
    for (int i = 0; i < n; i++) x |= (uint32_t)p[i] << (8 * i);
    return x;
}

int dab_wav_parse_header(const uint8_t *bytes, size_t n, uint64_t *out7) {
    size_t pos = 0;
#define NEED(k) do { if (n - pos < (size_t)(k)) return -1; } while (0)
    NEED(12);
    if (memcmp(bytes, "RIFF", 4) || memcmp(bytes + 8, "WAVE", 4)) return -1;
    pos = 12;
    NEED(24);
    const uint8_t *f = bytes + pos;
    pos += 24;
    if (memcmp(f, "fmt ", 4)) return -1;
    const uint32_t fmt_size = rd(f + 4, 4);
    if (fmt_size != 16 && fmt_size != 18 && fmt_size != 40) return -1;
    uint32_t code = rd(f + 8, 2);
    if (code != 1 && code != 3 && code != 6 && code != 7 && code != 0xFFFE) return -1;
    const uint32_t channels = rd(f + 10, 2);
    if (channels != 1 && channels != 2) return -1;
    const uint32_t rate = rd(f + 12, 4), bits = rd(f + 22, 2);
    if (fmt_size > 16) {
        const size_t ext = fmt_size - 16;
        NEED(ext);
        const uint8_t *e = bytes + pos;
        pos += ext;
        const uint32_t ext_size = rd(e, 2);
        if (ext_size != ext - 2) return -1;
        if (ext_size == 22) {
            const uint32_t sub = rd(e + 8, 2);
            if (sub != 1 && sub != 3 && sub != 6 && sub != 7) return -1;
            static const uint8_t GUID[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
            if (memcmp(GUID, e + 10, 14)) return -1;
            code = sub;
        }
    }
    if (code != 1) {
        NEED(8);
        if (memcmp(bytes + pos, "fact", 4)) return -1;
        const uint32_t fact = rd(bytes + pos + 4, 4);
        pos += 8;
        if (fact < 4) return -1;
        NEED(fact);
        pos += fact;
    }
    uint32_t data_size;
    for (;;) {
        NEED(8);
        const int is_data = memcmp(bytes + pos, "data", 4) == 0;
        const uint32_t size = rd(bytes + pos + 4, 4);
        pos += 8;
        if (is_data) { data_size = size; break; }
        NEED(size);
        pos += size;
    }
#undef NEED
    int fmt = -1;
    if (code == 1) fmt = bits == 8 ? 14 : bits == 16 ? 15 : bits == 24 ? 16 : bits == 32 ? 17 : -1;
    else if (code == 3) fmt = bits == 32 ? 18 : bits == 64 ? 19 : -1;
    else if (code == 6) fmt = bits == 8 ? 20 : -1;
    else if (code == 7) fmt = bits == 8 ? 21 : -1;
    if (fmt < 0) return -1;
    out7[0] = (uint64_t)fmt; out7[1] = code; out7[2] = channels; out7[3] = rate; out7[4] = bits;
    out7[5] = data_size; out7[6] = pos;
    return 0;
}
