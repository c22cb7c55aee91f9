/* dab_oracle_dabplus.c -- CPU oracle (TEST INFRASTRUCTURE, never shipped, never on the product path) for the DAB+
 * outer code: what AAC_Frame_Processor does between the channel decoder's bytes and the AAC access units
 * (src/dab/audio/aac_frame_processor.cpp:127-361) and the RS(120,110) decoding it relies on
 * (src/dab/algorithms/reed_solomon_decoder.cpp, a Berlekamp-Massey / Chien / Forney decoder), restated from
 * ETSI TS 102 563 clause 6 and the textbook algorithm; pinned against the reference's own objects compiled in place
 * (oracle/ref_harness_dabplus.cpp -> oracle/_ref) by tests/test_oracle_dabplus.py.
 */
#include <stdlib.h>
#include <string.h>

#include "dab_oracle.h"

/* ---- GF(2^8), p(x) = x^8 + x^4 + x^3 + x^2 + 1 (TS 102 563 6.1) ---- */
static uint8_t gf_exp[512];
static uint8_t gf_log[256];
static int gf_ready;

static void gf_init(void) {
    if (gf_ready) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        gf_exp[i] = (uint8_t)x; gf_exp[i + 255] = (uint8_t)x;
        gf_log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x11D;
    }
    gf_exp[510] = gf_exp[0]; gf_exp[511] = gf_exp[1];
    gf_log[0] = 0;
    gf_ready = 1;
}
static uint8_t gf_mul(uint8_t a, uint8_t b) { return (a && b) ? gf_exp[gf_log[a] + gf_log[b]] : 0; }
static uint8_t gf_div(uint8_t a, uint8_t b) { return a ? gf_exp[gf_log[a] + 255 - gf_log[b]] : 0; }     /* b != 0 */
static uint8_t gf_pow_alpha(int e) { e %= 255; if (e < 0) e += 255; return gf_exp[e]; }

#define RS_N 120
#define RS_ROOTS 10
#define RS_PAD 135

int dab_rs120_decode(uint8_t *cw, int *positions) {
    gf_init();
    /* syndromes S_i = cw(alpha^i), cw[0] the highest-order coefficient (:236-249) */
    uint8_t S[RS_ROOTS];
    int any = 0;
    for (int i = 0; i < RS_ROOTS; i++) {
        uint8_t s = 0;
        for (int j = 0; j < RS_N; j++) s = (uint8_t)(gf_mul(s, gf_pow_alpha(i)) ^ cw[j]);
        S[i] = s;
        any |= s;
    }
    if (!any) return 0;
    /* Berlekamp-Massey (:322-364): C = connection polynomial, B = previous, L = current LFSR length */
    uint8_t C[RS_ROOTS + 1] = {1}, B[RS_ROOTS + 1] = {1}, T[RS_ROOTS + 1];
    int L = 0;
    for (int r = 1; r <= RS_ROOTS; r++) {
        uint8_t d = 0;
        for (int i = 0; i < r; i++) d ^= gf_mul(C[i], S[r - 1 - i]);
        if (d == 0) {
            memmove(B + 1, B, RS_ROOTS); B[0] = 0;                       /* B <- x B */
        } else {
            T[0] = C[0];
            for (int i = 0; i < RS_ROOTS; i++) T[i + 1] = (uint8_t)(C[i + 1] ^ gf_mul(d, B[i]));    /* T <- C - d x B */
            if (2 * L <= r - 1) {
                L = r - L;
                for (int i = 0; i <= RS_ROOTS; i++) B[i] = gf_div(C[i], d);                          /* B <- C / d */
            } else {
                memmove(B + 1, B, RS_ROOTS); B[0] = 0;
            }
            memcpy(C, T, sizeof(C));
        }
    }
    int deg = 0;
    for (int i = 0; i <= RS_ROOTS; i++) if (C[i]) deg = i;
    /* Chien search over alpha^1 .. alpha^255; a root alpha^i marks symbol index i - 1 of the padded block (:374-399) */
    int root[RS_ROOTS], loc[RS_ROOTS], count = 0;
    for (int i = 1; i <= 255 && count < deg; i++) {
        uint8_t q = 1;
        for (int j = 1; j <= deg; j++) if (C[j]) q ^= gf_mul(C[j], gf_pow_alpha(i * j));
        if (q) continue;
        root[count] = i; loc[count] = i - 1; count++;
    }
    if (count != deg) return -1;                                         /* :401-408 */
    /* Forney (:414-479): omega = S C mod x^deg; e = omega(X^-1) X / C'(X^-1) with first root alpha^0 */
    uint8_t omega[RS_ROOTS];
    for (int i = 0; i < deg; i++) {
        uint8_t t = 0;
        for (int j = 0; j <= i; j++) t ^= gf_mul(S[i - j], C[j]);
        omega[i] = t;
    }
    for (int k = count - 1; k >= 0; k--) {
        uint8_t num1 = 0, den = 0;
        for (int i = deg - 1; i >= 0; i--) num1 ^= gf_mul(omega[i], gf_pow_alpha(i * root[k]));
        const uint8_t num2 = gf_pow_alpha(-root[k]);                     /* alpha^(root (fcr - 1)), fcr = 0 */
        const int top = ((deg < RS_ROOTS - 1) ? deg : (RS_ROOTS - 1)) & ~1;
        for (int i = top; i >= 0; i -= 2) den ^= gf_mul(C[i + 1], gf_pow_alpha(i * root[k]));
        if (num1 != 0 && loc[k] >= RS_PAD) cw[loc[k] - RS_PAD] ^= gf_div(gf_mul(num1, num2), den);
    }
    if (positions) for (int i = 0; i < count; i++) positions[i] = loc[i];
    return count;
}

void dab_rs120_encode(const uint8_t *data, uint8_t *parity) {
    gf_init();
    /* g(x) = prod_{i=0..9} (x + alpha^i); systematic: parity = data(x) x^10 mod g(x) */
    uint8_t g[RS_ROOTS + 1] = {1};
    for (int i = 0; i < RS_ROOTS; i++) {
        for (int j = i + 1; j > 0; j--) g[j] = (uint8_t)(g[j - 1] ^ gf_mul(g[j], gf_pow_alpha(i)));
        g[0] = gf_mul(g[0], gf_pow_alpha(i));
    }
    uint8_t rem[RS_ROOTS] = {0};
    for (int j = 0; j < RS_N - RS_ROOTS; j++) {
        const uint8_t fb = (uint8_t)(data[j] ^ rem[RS_ROOTS - 1]);
        for (int i = RS_ROOTS - 1; i > 0; i--) rem[i] = (uint8_t)(rem[i - 1] ^ gf_mul(fb, g[i]));
        rem[0] = gf_mul(fb, g[0]);
    }
    for (int i = 0; i < RS_ROOTS; i++) parity[i] = rem[RS_ROOTS - 1 - i];
}

static uint16_t crc16_msb(const uint8_t *x, size_t n, uint16_t poly, uint16_t init, uint16_t xorout) {
    uint16_t crc = init;
    for (size_t i = 0; i < n; i++) {
        crc ^= (uint16_t)((uint16_t)x[i] << 8);
        for (int j = 0; j < 8; j++) crc = (crc & 0x8000u) ? (uint16_t)((crc << 1) ^ poly) : (uint16_t)(crc << 1);
    }
    return (uint16_t)(crc ^ xorout);
}

uint16_t dab_firecode_crc(const uint8_t *data9) { return crc16_msb(data9, 9, 0x782F, 0x0000, 0x0000); }

/* ---- AAC_Frame_Processor ---- */
struct dab_aac_frame_processor {
    int wait_frame_start;       /* State::WAIT_FRAME_START */
    int curr_dab_frame;
    int prev_n;
    int synced;
    int desync_count;
    uint8_t *sf;
    size_t sf_cap;
};

dab_aac_frame_processor *dab_aac_create(void) {
    dab_aac_frame_processor *p = (dab_aac_frame_processor *)calloc(1, sizeof(*p));
    p->wait_frame_start = 1;
    return p;
}
void dab_aac_destroy(dab_aac_frame_processor *p) { if (p) { free(p->sf); free(p); } }

static int firecode_ok(const uint8_t *buf) {
    const uint16_t rx = (uint16_t)((buf[0] << 8) | buf[1]);
    return rx == dab_firecode_crc(buf + 2);
}

/* 12-bit big-endian fields packed back to back (read_au_start, :28-73) */
static int read_au_starts(const uint8_t *buf, int32_t *out, int count) {
    int bit = 0;
    for (int i = 0; i < count; i++) {
        int v = 0;
        for (int b = 0; b < 12; b++, bit++) v = (v << 1) | ((buf[bit >> 3] >> (7 - (bit & 7))) & 1);
        out[i] = v;
    }
    return (bit + 7) >> 3;
}

static void process_superframe(dab_aac_frame_processor *p, int n, dab_superframe_result *res) {
    const int n_rs = 5 * n / RS_N;
    uint8_t cw[RS_N];
    int pos[RS_ROOTS];
    for (int i = 0; i < n_rs; i++) {                                     /* ReedSolomonDecode :323-361 */
        for (int j = 0; j < RS_N; j++) cw[j] = p->sf[i + j * n_rs];
        const int cnt = dab_rs120_decode(cw, pos);
        if (cnt < 0) { res->rs_failed_index = i; p->desync_count++; return; }
        res->rs_corrected += cnt;
        for (int j = 0; j < cnt; j++) {
            const int k = pos[j] - RS_PAD;
            if (k < 0) continue;
            p->sf[i + k * n_rs] = cw[k];
        }
    }
    if (!firecode_ok(p->sf)) { p->desync_count++; return; }              /* :206-209 */
    res->firecode_ok = 1;
    p->desync_count = 0;
    p->synced = 1;
    const uint8_t descriptor = p->sf[2];
    const int dac_rate = (descriptor >> 6) & 1, sbr_flag = (descriptor >> 5) & 1;
    res->descriptor = descriptor;
    res->header_valid = 1;
    int num_aus = 0;
    if (!dac_rate && sbr_flag) num_aus = 2;
    if (dac_rate && sbr_flag) num_aus = 3;
    if (!dac_rate && !sbr_flag) num_aus = 4;
    if (dac_rate && !sbr_flag) num_aus = 6;
    res->num_aus = num_aus;
    const int nb_start_bytes = read_au_starts(p->sf + 3, &res->au_start[1], num_aus - 1);
    res->au_start[num_aus] = 110 * n_rs;
    res->au_start[0] = 3 + nb_start_bytes;
    const int sf_size = 5 * n;
    for (int i = 0; i < num_aus; i++) {                                  /* :286-317 */
        const int nb_au = res->au_start[i + 1] - res->au_start[i];
        const int nb_data = nb_au - 2;
        if (nb_data < 0 || res->au_start[i + 1] >= sf_size) { res->au_walk_stopped_at = i; return; }
        const uint8_t *au = p->sf + res->au_start[i];
        const uint16_t rx = (uint16_t)((au[nb_data] << 8) | au[nb_data + 1]);
        if (rx == crc16_msb(au, (size_t)nb_data, 0x1021, 0xFFFF, 0xFFFF)) res->au_crc_ok_mask |= 1u << i;
    }
}

int dab_aac_process(dab_aac_frame_processor *p, const uint8_t *frame, int n, dab_superframe_result *res, uint8_t *sf_out) {
    memset(res, 0, sizeof(*res));
    res->rs_failed_index = -1;
    res->au_walk_stopped_at = -1;
    if (n == 0 || n < 11) return -1;                                     /* :129-137 */
    if (p->prev_n != n) {                                                /* :140-147 */
        p->prev_n = n;
        if ((size_t)(5 * n) > p->sf_cap) { p->sf = (uint8_t *)realloc(p->sf, (size_t)(5 * n)); p->sf_cap = (size_t)(5 * n); }
        memset(p->sf, 0, (size_t)(5 * n));
        p->curr_dab_frame = 0;
        p->wait_frame_start = 1;
    }
    if (p->desync_count >= 10) { p->desync_count = 0; p->synced = 0; }   /* :151-154 */
    if (p->synced) p->wait_frame_start = 0;                              /* :158-160 */
    if (p->wait_frame_start) {
        if (!firecode_ok(frame)) { res->firecode_wait_failed = 1; return 0; }
        p->wait_frame_start = 0;
    }
    memcpy(p->sf + (size_t)p->curr_dab_frame * (size_t)n, frame, (size_t)n);
    p->curr_dab_frame++;
    if (p->curr_dab_frame == 5) {
        res->superframe_done = 1;
        process_superframe(p, n, res);
        if (sf_out) memcpy(sf_out, p->sf, (size_t)(5 * n));
        p->wait_frame_start = 1;
        p->curr_dab_frame = 0;
    }
    return 0;
}
