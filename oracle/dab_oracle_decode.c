/*
 * dab_oracle_decode.c -- CPU ORACLE (test infrastructure, NOT product code): channel-decode half
 * and the transmit-side vector generator.  See dab_oracle.h for scope.
 * Paths cited are relative to /root/reference.
 */
#include "dab_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <assert.h>

/* ------------------------------------------------------------------------------------------ */
/* constants                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* ETSI EN 300 401 table 13 in kept-count form: PI_n keeps 8+n of every 32 mother bits;
 * same data as src/dab/constants/puncture_codes.h:42-67 (derived here from the ETSI rule: the
 * n-th extra kept bit goes to 4-bit group order {0,4,2,6,1,5,3,7} cyclically, 8 per level) */
static uint8_t g_pi[24][8];
static int g_pi_ready = 0;
static const uint8_t PI_X_CODE[6] = { 2, 2, 2, 2, 2, 2 };      /* puncture_codes.h:69 */

static void ensure_pi(void) {
    if (g_pi_ready) return;
    static const int order[8] = { 0, 4, 2, 6, 1, 5, 3, 7 };
    for (int n = 1; n <= 24; n++) {
        int cnt[8];
        for (int g = 0; g < 8; g++) cnt[g] = 1;
        for (int e = 0; e < n; e++) cnt[order[e % 8]] += 1;
        for (int g = 0; g < 8; g++) g_pi[n - 1][g] = (uint8_t)cnt[g];
    }
    g_pi_ready = 1;
}

const uint8_t *dab_puncture_code(int pi) { ensure_pi(); assert(pi >= 1 && pi <= 24); return g_pi[pi - 1]; }
const uint8_t *dab_puncture_code_tail(void) { return PI_X_CODE; }

/* src/dab/algorithms/additive_scrambler.h:16-35, syncword 0xFFFF */
void dab_scrambler_bytes(uint8_t *out, size_t n) {
    uint16_t reg = 0xFFFF;
    for (size_t k = 0; k < n; k++) {
        uint8_t b = 0;
        for (int i = 0; i < 8; i++) {
            const uint8_t v = (uint8_t)(((reg >> 8) ^ (reg >> 4)) & 1u);
            b |= (uint8_t)(v << (7 - i));
            reg = (uint16_t)((reg << 1) | v);
        }
        out[k] = b;
    }
}

/* src/dab/algorithms/crc.h:25-68 with fic_decoder.cpp:19-31 parameters */
uint16_t dab_crc16(const uint8_t *x, size_t n) {
    uint16_t crc = 0xFFFF;
    for (size_t i = 0; i < n; i++) {
        crc ^= (uint16_t)((uint16_t)x[i] << 8);
        for (int j = 0; j < 8; j++)
            crc = (crc & 0x8000u) ? (uint16_t)((crc << 1) ^ 0x1021u) : (uint16_t)(crc << 1);
    }
    return (uint16_t)(crc ^ 0xFFFFu);
}

/* ------------------------------------------------------------------------------------------ */
/* Viterbi                                                                                      */
/* ------------------------------------------------------------------------------------------ */

#define VK 7
#define VR 4
#define VSTATES 64
static const uint8_t V_POLY[VR] = { 109, 79, 83, 109 };            /* dab_viterbi_decoder.cpp:25 */
/* dab_viterbi_decoder.cpp:31-41 */
#define V_MAX_ERROR      1016u                                     /* (127-(-127))*4 */
#define V_START_ERROR    0u
#define V_NONSTART_ERROR 5080u                                     /* 5*1016 */
#define V_RENORM_THRESH  60455u                                    /* 65535-5080 */

struct dab_viterbi {
    size_t traceback_length;
    size_t max_steps;
    int tie_rule;
    uint16_t metric[2][VSTATES];
    int cur;                          /* index of "old" metrics */
    uint64_t *decisions;              /* one word per trellis step */
    size_t current_decoded_bit;
    uint64_t accumulated_error;
    int16_t *depunctured;
    size_t depunctured_cap;
    int16_t branch[VR][VSTATES / 2];  /* ViterbiBranchTable: +127 if parity((2s)&poly) else -127 */
};

static int parity8(unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return (int)(v & 1u); }

dab_viterbi *dab_viterbi_create(size_t traceback_length, int tie_rule) {
    dab_viterbi *v = (dab_viterbi *)calloc(1, sizeof(*v));
    v->traceback_length = traceback_length;
    v->max_steps = traceback_length + (VK - 1);
    v->tie_rule = tie_rule;
    v->decisions = (uint64_t *)calloc(v->max_steps, sizeof(uint64_t));
    for (int r = 0; r < VR; r++)
        for (int s = 0; s < VSTATES / 2; s++)
            v->branch[r][s] = parity8((unsigned)(2 * s) & V_POLY[r]) ? 127 : -127;
    dab_viterbi_reset(v, 0);
    return v;
}

void dab_viterbi_destroy(dab_viterbi *v) { if (!v) return; free(v->decisions); free(v->depunctured); free(v); }

/* dab_viterbi_decoder.cpp:109-112 */
void dab_viterbi_reset(dab_viterbi *v, size_t starting_state) {
    v->cur = 0;
    for (int s = 0; s < VSTATES; s++) v->metric[0][s] = (uint16_t)V_NONSTART_ERROR;
    v->metric[0][starting_state % VSTATES] = (uint16_t)V_START_ERROR;
    v->current_decoded_bit = 0;
    v->accumulated_error = 0;
    memset(v->decisions, 0, v->max_steps * sizeof(uint64_t));
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
#include <immintrin.h>
/* The same trellis step sixteen butterflies per vector (what an upstream build does with its AVX2 core, so that bench.py's cpu_baseline_full
 * does not time a scalar port of it): every vector operation is the lane-wise form of a statement of the scalar loop below -- u16 wrapping
 * adds, the unsigned minimum, the decision as the scalar tie rule states it ((m1 <= m0) <=> min == m1; (m0 > m1) <=> !(min == m0)), the same
 * renormalisation -- so metrics, decision words and the accumulated error are identical (tests/test_oracle_properties.py compares the two
 * through DAB_ORACLE_SCALAR_VITERBI=1, incl. u16 wrap and renormalisation cases). */
__attribute__((target("avx2,bmi2")))
static uint64_t viterbi_steps_avx2(dab_viterbi *v, const int16_t *sym, size_t n_sym) {
    uint64_t total = 0;
    __m256i br[VR][2];
    for (int r = 0; r < VR; r++) for (int h = 0; h < 2; h++) br[r][h] = _mm256_loadu_si256((const __m256i *)&v->branch[r][16 * h]);
    const __m256i kmax = _mm256_set1_epi16((short)V_MAX_ERROR);
    for (size_t s0 = 0; s0 < n_sym; s0 += VR) {
        assert(v->current_decoded_bit < v->max_steps);
        const uint16_t *old = v->metric[v->cur];
        uint16_t *nw = v->metric[v->cur ^ 1];
        uint32_t D0 = 0, D1 = 0;                                    /* bit s: decision of new state 2s / 2s + 1 */
        for (int h = 0; h < 2; h++) {                               /* butterflies 16 h .. 16 h + 15 */
            __m256i e = _mm256_setzero_si256();
            for (int r = 0; r < VR; r++)
                e = _mm256_add_epi16(e, _mm256_abs_epi16(_mm256_sub_epi16(br[r][h], _mm256_set1_epi16(sym[s0 + r]))));
            const __m256i m = _mm256_sub_epi16(kmax, e);
            const __m256i lo = _mm256_loadu_si256((const __m256i *)(old + 16 * h)), hi = _mm256_loadu_si256((const __m256i *)(old + 32 + 16 * h));
            /* core model 1 (the upstream SIMD cores): _mm*_adds_epu16, the candidates SATURATE at 65535; model 0 (the scalar core): they wrap */
            const __m256i m0 = v->tie_rule ? _mm256_adds_epu16(lo, e) : _mm256_add_epi16(lo, e), m1 = v->tie_rule ? _mm256_adds_epu16(hi, m) : _mm256_add_epi16(hi, m);
            const __m256i m2 = v->tie_rule ? _mm256_adds_epu16(lo, m) : _mm256_add_epi16(lo, m), m3 = v->tie_rule ? _mm256_adds_epu16(hi, e) : _mm256_add_epi16(hi, e);
            const __m256i n0 = _mm256_min_epu16(m0, m1), n1 = _mm256_min_epu16(m2, m3);
            uint32_t k0, k1;
            if (v->tie_rule) {
                k0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi16(n0, m1));
                k1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi16(n1, m3));
            } else {
                k0 = ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi16(n0, m0));
                k1 = ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi16(n1, m2));
            }
            D0 |= (uint32_t)_pext_u32(k0, 0xAAAAAAAAu) << (16 * h);
            D1 |= (uint32_t)_pext_u32(k1, 0xAAAAAAAAu) << (16 * h);
            /* new states 2s, 2s + 1 interleaved: nw[32 h .. 32 h + 31] */
            const __m256i il = _mm256_unpacklo_epi16(n0, n1), ih = _mm256_unpackhi_epi16(n0, n1);
            _mm256_storeu_si256((__m256i *)(nw + 32 * h), _mm256_permute2x128_si256(il, ih, 0x20));
            _mm256_storeu_si256((__m256i *)(nw + 32 * h + 16), _mm256_permute2x128_si256(il, ih, 0x31));
        }
        v->decisions[v->current_decoded_bit] = _pdep_u64((uint64_t)D0, 0x5555555555555555ull) | _pdep_u64((uint64_t)D1, 0xAAAAAAAAAAAAAAAAull);
        if (nw[0] >= V_RENORM_THRESH) {
            uint16_t mn = nw[0];
            for (int s = 1; s < VSTATES; s++) if (nw[s] < mn) mn = nw[s];
            for (int s = 0; s < VSTATES; s++) nw[s] = (uint16_t)(nw[s] - mn);
            total += mn;
        }
        v->cur ^= 1;
        v->current_decoded_bit++;
    }
    return total;
}
static int g_scalar_viterbi = -1;
#endif

/* published ViterbiDecoderCpp scalar core: one trellis step = 32 butterflies on u16 metrics */
static uint64_t viterbi_steps(dab_viterbi *v, const int16_t *sym, size_t n_sym) {
#if defined(__x86_64__) && defined(__GNUC__) && !defined(DAB_ORACLE_NO_CLONES)
    if (g_scalar_viterbi < 0) {
        const char *e = getenv("DAB_ORACLE_SCALAR_VITERBI");
        g_scalar_viterbi = (e && e[0] == '1') || !(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2"));
    }
    if (!g_scalar_viterbi) return viterbi_steps_avx2(v, sym, n_sym);
#endif
    uint64_t total = 0;
    for (size_t s0 = 0; s0 < n_sym; s0 += VR) {
        assert(v->current_decoded_bit < v->max_steps);
        const uint16_t *old = v->metric[v->cur];
        uint16_t *nw = v->metric[v->cur ^ 1];
        uint64_t dec = 0;
        for (int s = 0; s < VSTATES / 2; s++) {
            uint16_t e = 0;
            for (int r = 0; r < VR; r++) {
                const int16_t d = (int16_t)(v->branch[r][s] - sym[s0 + r]);
                e = (uint16_t)(e + (uint16_t)(d < 0 ? -d : d));
            }
            const uint16_t m  = (uint16_t)(V_MAX_ERROR - e);
            /* core model 0: uint16_t arithmetic of the scalar core, the sum wraps; model 1: adds_epu16 of the SIMD cores, it saturates */
#define V_ADD(a, b) (v->tie_rule ? (uint16_t)((unsigned)(a) + (unsigned)(b) > 65535u ? 65535u : (unsigned)(a) + (unsigned)(b)) : (uint16_t)((a) + (b)))
            const uint16_t m0 = V_ADD(old[s], e);
            const uint16_t m1 = V_ADD(old[s + VSTATES / 2], m);
            const uint16_t m2 = V_ADD(old[s], m);
            const uint16_t m3 = V_ADD(old[s + VSTATES / 2], e);
#undef V_ADD
            const int d0 = v->tie_rule ? (m1 <= m0) : (m0 > m1);
            const int d1 = v->tie_rule ? (m3 <= m2) : (m2 > m3);
            nw[2 * s]     = d0 ? m1 : m0;
            nw[2 * s + 1] = d1 ? m3 : m2;
            dec |= ((uint64_t)d0 << (2 * s)) | ((uint64_t)d1 << (2 * s + 1));
        }
        v->decisions[v->current_decoded_bit] = dec;
        if (nw[0] >= V_RENORM_THRESH) {
            uint16_t mn = nw[0];
            for (int s = 1; s < VSTATES; s++) if (nw[s] < mn) mn = nw[s];
            for (int s = 0; s < VSTATES; s++) nw[s] = (uint16_t)(nw[s] - mn);
            total += mn;
        }
        v->cur ^= 1;
        v->current_decoded_bit++;
    }
    return total;
}

/* dab_viterbi_decoder.cpp:114-181 */
size_t dab_viterbi_update(dab_viterbi *v, const int8_t *punctured, size_t n_punctured,
                          const uint8_t *code, size_t n_code, size_t requested) {
    assert(requested % VR == 0);
    if (requested > v->depunctured_cap) {
        v->depunctured = (int16_t *)realloc(v->depunctured, requested * sizeof(int16_t));
        v->depunctured_cap = requested;
    }
    size_t ip = 0, ic = 0, io = 0;
    while (io < requested) {
        const size_t keep = code[ic];
        if (n_punctured - ip < keep) return 0;                     /* :157-160: res is returned as initialised, {0, 0}: no step runs, 0 consumed */
        for (size_t i = 0; i < keep; i++) v->depunctured[io++] = (int16_t)punctured[ip++];
        for (size_t i = keep; i < VR; i++) v->depunctured[io++] = 0;
        ic = (ic + 1) % n_code;
    }
    v->accumulated_error += viterbi_steps(v, v->depunctured, io);
    return ip;
}

/* dab_viterbi_decoder.cpp:124-129 over the published core chainback (Karn layout, MSB-first bytes) */
uint64_t dab_viterbi_chainback(dab_viterbi *v, uint8_t *bytes_out, size_t n_bytes, size_t end_state) {
    const size_t total_bits = n_bytes * 8u;
    unsigned reg = (unsigned)((end_state % VSTATES) << 2);           /* 8-bit window, state in bits 7..2 */
    for (size_t i = 0; i < total_bits; i++) {
        const size_t bit = (total_bits - 1) - i;
        const uint64_t dec = v->decisions[bit + (VK - 1)];
        const unsigned in = (unsigned)((dec >> (reg >> 2)) & 1u);
        reg = (reg >> 1) | (in << 7);
        bytes_out[bit / 8] = (uint8_t)reg;
    }
    return v->accumulated_error + (uint64_t)v->metric[v->cur][end_state % VSTATES];
}

size_t dab_viterbi_current_decoded_bit(const dab_viterbi *v) { return v->current_decoded_bit; }
const uint64_t *dab_viterbi_decisions(const dab_viterbi *v) { return v->decisions; }
const uint16_t *dab_viterbi_metrics(const dab_viterbi *v) { return v->metric[v->cur]; }

/* ------------------------------------------------------------------------------------------ */
/* FIC                                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* src/dab/fic/fic_decoder.cpp:53-117 */
uint64_t dab_fic_decode_group(const int8_t *bits, int tie_rule, uint8_t *bytes, uint32_t *crc_ok_mask) {
    dab_viterbi *v = dab_viterbi_create(768, tie_rule);              /* :45 traceback = nb_encoded_bits/3 */
    size_t pos = 0, n = DAB_NB_FIB_GROUP_BITS;
    pos += dab_viterbi_update(v, bits + pos, n - pos, dab_puncture_code(16), 8, 128 * 21);   /* :78 */
    pos += dab_viterbi_update(v, bits + pos, n - pos, dab_puncture_code(15), 8, 128 * 3);    /* :80 */
    pos += dab_viterbi_update(v, bits + pos, n - pos, PI_X_CODE, 6, 24);                     /* :82 */
    assert(pos == n);
    const uint64_t err = dab_viterbi_chainback(v, bytes, 96, 0);     /* :87 */
    uint8_t prbs[96];
    dab_scrambler_bytes(prbs, 96);                                   /* :91-95 */
    for (int i = 0; i < 96; i++) bytes[i] ^= prbs[i];
    uint32_t mask = 0;
    for (int i = 0; i < 3; i++) {                                    /* :103-116 */
        const uint8_t *fib = bytes + 32 * i;
        const uint16_t rx = (uint16_t)((fib[30] << 8) | fib[31]);
        if (rx == dab_crc16(fib, 30)) mask |= 1u << i;
    }
    if (crc_ok_mask) *crc_ok_mask = mask;
    dab_viterbi_destroy(v);
    return err;
}

/* ------------------------------------------------------------------------------------------ */
/* MSC                                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* ETSI EN 300 401 tables 8+15 {size CU, kbps, level, L1..L4, PI1..PI4, padding};
 * same data as src/dab/constants/subchannel_protection_tables.h:21-86 */
static const uint16_t UEP_TABLE[64][12] = {
    {16,32,5,3,4,17,0,5,3,2,0,0},       {21,32,4,3,3,18,0,11,6,5,0,0},      {24,32,3,3,4,14,3,15,9,6,8,0},
    {29,32,2,3,4,14,3,22,13,8,13,0},    {35,32,1,3,5,13,3,24,17,12,17,4},   {24,48,5,4,3,26,3,5,4,2,3,0},
    {29,48,4,3,4,26,3,9,6,4,6,0},       {35,48,3,3,4,26,3,15,10,6,9,4},     {42,48,2,3,4,26,3,24,14,8,15,0},
    {52,48,1,3,5,25,3,24,18,13,18,0},   {29,56,5,6,10,23,3,5,4,2,3,0},      {35,56,4,6,10,23,3,9,6,4,5,0},
    {42,56,3,6,12,21,3,16,7,6,9,0},     {52,56,2,6,10,23,3,23,13,8,13,8},   {32,64,5,6,9,31,2,5,3,2,3,0},
    {42,64,4,6,9,33,0,11,6,5,0,0},      {48,64,3,6,12,27,3,16,8,6,9,0},     {58,64,2,6,10,29,3,23,13,8,13,8},
    {70,64,1,6,11,28,3,24,18,12,18,4},  {40,80,5,6,10,41,3,6,3,2,3,0},      {52,80,4,6,10,41,3,11,6,5,6,0},
    {58,80,3,6,11,40,3,16,8,6,7,0},     {70,80,2,6,10,41,3,23,13,8,13,8},   {84,80,1,6,10,41,3,24,17,12,18,4},
    {48,96,5,7,9,53,3,5,4,2,4,0},       {58,96,4,7,10,52,3,9,6,4,6,0},      {70,96,3,6,12,51,3,16,9,6,10,4},
    {84,96,2,6,10,53,3,22,12,9,12,0},   {104,96,1,6,13,50,3,24,18,13,19,0}, {58,112,5,14,17,50,3,5,4,2,5,0},
    {70,112,4,11,21,49,3,9,6,4,8,0},    {84,112,3,11,23,47,3,16,8,6,9,0},   {104,112,2,11,21,49,3,23,12,9,14,4},
    {84,128,5,12,19,62,3,5,3,2,4,0},    {64,128,4,11,21,61,3,11,6,5,7,0},   {96,128,3,11,22,60,3,16,9,6,10,4},
    {116,128,2,11,21,61,3,22,12,9,14,0},{140,128,1,11,20,62,3,24,17,13,19,8},{80,160,5,11,19,87,3,5,4,2,4,0},
    {104,160,4,11,23,83,3,11,6,5,9,0},  {116,160,3,11,24,82,3,16,8,6,11,0}, {140,160,2,11,21,85,3,22,11,9,13,0},
    {168,160,1,11,22,84,3,24,18,12,19,0},{96,192,5,11,20,110,3,6,4,2,5,0},  {116,192,4,11,22,108,3,10,6,4,9,0},
    {140,192,3,11,24,106,3,16,10,6,11,0},{168,192,2,11,20,110,3,22,13,9,13,8},{208,192,1,11,21,109,3,24,20,13,24,0},
    {116,224,5,12,22,131,3,8,6,2,6,4},  {140,224,4,12,26,127,3,12,8,4,11,0},{168,224,3,11,20,134,3,16,10,7,9,0},
    {208,224,2,11,22,132,3,24,16,10,15,0},{232,224,1,11,24,130,3,24,20,12,20,4},{128,256,5,11,24,154,3,6,5,2,5,0},
    {168,256,4,11,24,154,3,12,9,5,10,4},{192,256,3,11,27,151,3,16,10,7,10,0},{232,256,2,11,22,156,3,24,14,10,13,8},
    {280,256,1,11,26,152,3,24,19,14,18,4},{160,320,5,11,26,200,3,8,5,2,6,4}, {208,320,4,11,25,201,3,13,9,5,10,8},
    {280,320,2,11,26,200,3,24,17,9,17,0},{192,384,5,11,27,247,3,8,6,2,7,0}, {280,384,3,11,24,250,3,16,9,7,10,4},
    {416,384,1,12,28,245,3,24,20,14,23,8},
};

/* ETSI EN 300 401 tables 9/18 and 10/20: {CU multiple, m1,b1, m2,b2, PI1, PI2};
 * same data as subchannel_protection_tables.h:121-139 */
static const int EEP_A[4][7] = { {12,6,-3,0,3,24,23}, {8,2,-3,4,3,14,13}, {6,6,-3,0,3,8,7}, {4,4,-3,2,3,3,2} };
static const int EEP_2A_SPECIAL[7] = { 8,0,5,0,1,13,12 };
static const int EEP_B[4][7] = { {27,24,-3,0,3,10,9}, {21,24,-3,0,3,6,5}, {18,24,-3,0,3,4,3}, {15,24,-3,0,3,2,1} };

/* GetEEPDescriptor subchannel_protection_tables.h:145-154 ; DecodeEEP msc_decoder.cpp:77-94 ; DecodeUEP :118-137 */
int dab_subchannel_plan(const dab_subchannel *sc, int *pi, int *lx, int *n_decoded_bytes) {
    int nseg = 0, total_lx = 0;
    if (!sc->is_uep) {
        const int *d;
        if (sc->eep_type == 0) d = (sc->length == 8) ? EEP_2A_SPECIAL : EEP_A[sc->eep_prot_level];
        else d = EEP_B[sc->eep_prot_level];
        const int n = sc->length / d[0];
        pi[0] = d[5]; lx[0] = d[1] * n + d[2];
        pi[1] = d[6]; lx[1] = d[3] * n + d[4];
        nseg = 2;
    } else {
        const uint16_t *d = UEP_TABLE[sc->uep_prot_index];
        for (int i = 0; i < 4; i++) { lx[i] = d[3 + i]; pi[i] = d[7 + i]; }
        nseg = 4;
    }
    for (int i = 0; i < nseg; i++) total_lx += lx[i];
    if (n_decoded_bytes) *n_decoded_bytes = 4 * total_lx;           /* (32*sum(Lx)+6-6)/8, msc_decoder.cpp:99-103 */
    return nseg;
}

uint64_t dab_msc_decode_logical(const dab_subchannel *sc, const int8_t *bits, int tie_rule, uint8_t *bytes, int *n_bytes) {
    int pi[4], lx[4], nb = 0;
    const int nseg = dab_subchannel_plan(sc, pi, lx, &nb);
    const size_t n = (size_t)sc->length * 64;
    dab_viterbi *v = dab_viterbi_create(n, tie_rule);                /* msc_decoder.cpp:37 */
    size_t pos = 0;
    for (int i = 0; i < nseg; i++) {
        if (lx[i] == 0) continue;   /* update() with 0 requested symbols consumes nothing */
        pos += dab_viterbi_update(v, bits + pos, n - pos, dab_puncture_code(pi[i]), 8, (size_t)128 * lx[i]);
    }
    pos += dab_viterbi_update(v, bits + pos, n - pos, PI_X_CODE, 6, 24);
    const int decoded_bits = (int)dab_viterbi_current_decoded_bit(v) - 6;
    const int decoded_bytes = decoded_bits / 8;
    const uint64_t err = dab_viterbi_chainback(v, bytes, (size_t)decoded_bytes, 0);
    uint8_t *prbs = (uint8_t *)malloc((size_t)decoded_bytes);
    dab_scrambler_bytes(prbs, (size_t)decoded_bytes);
    for (int i = 0; i < decoded_bytes; i++) bytes[i] ^= prbs[i];
    free(prbs);
    if (n_bytes) *n_bytes = decoded_bytes;
    dab_viterbi_destroy(v);
    (void)nb;
    return err;
}

/* src/dab/msc/cif_deinterleaver.cpp:13-71 */
struct dab_deinterleaver { int nb_bits; int curr; int stored; int8_t *buf; };
static const int CIF_OFFSETS[16] = { 0,8,4,12, 2,10,6,14, 1,9,5,13, 3,11,7,15 };

dab_deinterleaver *dab_deinterleaver_create(int nb_bytes) {
    dab_deinterleaver *d = (dab_deinterleaver *)calloc(1, sizeof(*d));
    d->nb_bits = nb_bytes * 8;
    d->buf = (int8_t *)calloc((size_t)d->nb_bits * 16, 1);
    return d;
}
void dab_deinterleaver_destroy(dab_deinterleaver *d) { if (!d) return; free(d->buf); free(d); }
void dab_deinterleaver_consume(dab_deinterleaver *d, const int8_t *bits) {
    memcpy(d->buf + (size_t)d->nb_bits * d->curr, bits, (size_t)d->nb_bits);
    d->curr = (d->curr + 1) % 16;
    if (d->stored < 16) d->stored++;
}
int dab_deinterleaver_deinterleave(dab_deinterleaver *d, int8_t *out) {
    if (d->stored < 16) return 0;
    for (int i = 0; i < d->nb_bits; i++) {
        const int age = 15 - CIF_OFFSETS[i % 16];                    /* :64-66 */
        const int frame = ((d->curr - 1) - age + 32) % 16;           /* :48-51 */
        out[i] = d->buf[(size_t)frame * d->nb_bits + i];
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* transmit side (vector generator)                                                             */
/* ------------------------------------------------------------------------------------------ */

void dab_conv_encode(const uint8_t *bytes, size_t n_bits, uint8_t *mother) {
    unsigned sr = 0;
    for (size_t i = 0; i < n_bits + 6; i++) {
        const unsigned bit = (i < n_bits) ? ((bytes[i / 8] >> (7 - (i % 8))) & 1u) : 0u;
        sr = ((sr << 1) | bit) & 0x7Fu;
        for (int r = 0; r < VR; r++) mother[4 * i + r] = (uint8_t)parity8(sr & V_POLY[r]);
    }
}

size_t dab_puncture(const uint8_t *mother, size_t n_mother, const uint8_t *code, size_t n_code, uint8_t *out) {
    size_t io = 0, ic = 0;
    for (size_t i = 0; i < n_mother; i += VR) {
        for (size_t k = 0; k < code[ic]; k++) out[io++] = mother[i + k];
        ic = (ic + 1) % n_code;
    }
    return io;
}

void dab_fic_encode_group(const uint8_t *fib_data, uint8_t *out_bits) {
    uint8_t bytes[96], prbs[96], mother[4 * (768 + 6)];
    for (int i = 0; i < 3; i++) {
        memcpy(bytes + 32 * i, fib_data + 30 * i, 30);
        const uint16_t crc = dab_crc16(fib_data + 30 * i, 30);
        bytes[32 * i + 30] = (uint8_t)(crc >> 8);
        bytes[32 * i + 31] = (uint8_t)(crc & 0xFF);
    }
    dab_scrambler_bytes(prbs, 96);
    for (int i = 0; i < 96; i++) bytes[i] ^= prbs[i];
    dab_conv_encode(bytes, 768, mother);
    size_t o = 0;
    o += dab_puncture(mother, 128 * 21, dab_puncture_code(16), 8, out_bits + o);
    o += dab_puncture(mother + 128 * 21, 128 * 3, dab_puncture_code(15), 8, out_bits + o);
    o += dab_puncture(mother + 128 * 24, 24, PI_X_CODE, 6, out_bits + o);
    assert(o == DAB_NB_FIB_GROUP_BITS);
}

void dab_msc_encode_logical(const dab_subchannel *sc, const uint8_t *bytes_in, uint8_t *out_bits) {
    int pi[4], lx[4], nb = 0;
    const int nseg = dab_subchannel_plan(sc, pi, lx, &nb);
    uint8_t *bytes = (uint8_t *)malloc((size_t)nb);
    uint8_t *mother = (uint8_t *)malloc((size_t)4 * ((size_t)nb * 8 + 6));
    dab_scrambler_bytes(bytes, (size_t)nb);
    for (int i = 0; i < nb; i++) bytes[i] ^= bytes_in[i];
    dab_conv_encode(bytes, (size_t)nb * 8, mother);
    size_t o = 0, m = 0;
    for (int i = 0; i < nseg; i++) {
        if (lx[i] == 0) continue;
        o += dab_puncture(mother + m, (size_t)128 * lx[i], dab_puncture_code(pi[i]), 8, out_bits + o);
        m += (size_t)128 * lx[i];
    }
    o += dab_puncture(mother + m, 24, PI_X_CODE, 6, out_bits + o);
    /* UEP rows carry padding bits (zeros) up to the sub-channel size */
    const size_t total = (size_t)sc->length * 64;
    assert(o <= total);
    while (o < total) out_bits[o++] = 0;
    free(bytes); free(mother);
}

/* bits (0/1, RX frame-bit layout) -> time domain, NULL first */
void dab_modulate_frame(const uint8_t *frame_bits, const int *mapper, dab_cf32 *out) {
    static dab_cf32 prs[DAB_NB_FFT];
    dab_cf32 last[DAB_NB_FFT], cur[DAB_NB_FFT];
    dab_get_prs_fft(prs);
    const float A = 0.707106769084930420f;
    for (int i = 0; i < DAB_NB_NULL_PERIOD; i++) { out[i].re = 0.0f; out[i].im = 0.0f; }
    dab_cf32 *p = out + DAB_NB_NULL_PERIOD;
    dab_ifft2048(prs, p + DAB_NB_CYCLIC_PREFIX);
    memcpy(p, p + DAB_NB_FFT, sizeof(dab_cf32) * DAB_NB_CYCLIC_PREFIX);
    memcpy(last, prs, sizeof(prs));
    for (int s = 0; s < DAB_NB_FRAME_SYMBOLS - 1; s++) {
        p += DAB_NB_SYMBOL_PERIOD;
        const uint8_t *b = frame_bits + (size_t)s * DAB_NB_SYM_BITS;
        memset(cur, 0, sizeof(cur));
        for (int n = 0; n < DAB_NB_DATA_CARRIERS; n++) {
            const int c = mapper[n];
            const int k = (c < 768) ? (c - 768) : (c - 768 + 1);
            const int bin = (DAB_NB_FFT + k) % DAB_NB_FFT;
            /* QPSK z = ((1-2 p_n) + j(1-2 p_{n+1536}))/sqrt2 ; differential X_{s+1} = X_s * z */
            const float zr = b[n] ? -A : A, zi = b[n + DAB_NB_DATA_CARRIERS] ? -A : A;
            const dab_cf32 x = last[bin];
            cur[bin].re = x.re * zr - x.im * zi;
            cur[bin].im = x.re * zi + x.im * zr;
        }
        dab_ifft2048(cur, p + DAB_NB_CYCLIC_PREFIX);
        memcpy(p, p + DAB_NB_FFT, sizeof(dab_cf32) * DAB_NB_CYCLIC_PREFIX);
        memcpy(last, cur, sizeof(cur));
    }
}

/* src/ofdm/ofdm_modulator.cpp:49-156 */
void dab_modulate_frame_reference_payload(const uint8_t *payload, dab_cf32 *out) {
    static dab_cf32 prs[DAB_NB_FFT];
    dab_cf32 last[DAB_NB_FFT], cur[DAB_NB_FFT];
    dab_get_prs_fft(prs);
    const float A = 1.0f / 1.41421356237309505f;
    const dab_cf32 MAP[4] = { {-A,-A}, {A,-A}, {A,A}, {-A,A} };      /* :102-103 */
    for (int i = 0; i < DAB_NB_NULL_PERIOD; i++) { out[i].re = 0.0f; out[i].im = 0.0f; }
    dab_cf32 *p = out + DAB_NB_NULL_PERIOD;
    dab_ifft2048(prs, p + DAB_NB_CYCLIC_PREFIX);
    memcpy(p, p + DAB_NB_FFT, sizeof(dab_cf32) * DAB_NB_CYCLIC_PREFIX);
    memcpy(last, prs, sizeof(prs));
    for (int s = 0; s < DAB_NB_FRAME_SYMBOLS - 1; s++) {
        p += DAB_NB_SYMBOL_PERIOD;
        const uint8_t *d = payload + (size_t)s * 384;
        memset(cur, 0, sizeof(cur));
        for (int c = 0; c < DAB_NB_DATA_CARRIERS; c++) {              /* natural carrier order :106-126 */
            const int bin = (c < 768) ? (DAB_NB_FFT - 768 + c) : (c - 768 + 1);
            const dab_cf32 z = MAP[(d[c / 4] >> (2 * (c % 4))) & 3];
            const dab_cf32 x = last[bin];
            cur[bin].re = x.re * z.re - x.im * z.im;                  /* :133-143 */
            cur[bin].im = x.re * z.im + x.im * z.re;
        }
        dab_ifft2048(cur, p + DAB_NB_CYCLIC_PREFIX);
        memcpy(p, p + DAB_NB_FFT, sizeof(dab_cf32) * DAB_NB_CYCLIC_PREFIX);
        memcpy(last, cur, sizeof(cur));
    }
}
