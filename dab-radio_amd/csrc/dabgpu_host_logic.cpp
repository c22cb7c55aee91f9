// dabgpu_host_logic.cpp -- see dabgpu_host_logic.h.  No device code and no HIP call in this file.
#include "dabgpu_host_logic.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <string>

static thread_local std::string g_last_error;

void dabgpu_set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_last_error = buf;
}


extern "C" {

const char* dabgpu_strerror(int status) {
    switch (status) {
    case DABGPU_OK: return "ok";
    case DABGPU_ERR_NO_DEVICE: return "no usable gfx950 device (this library has no CPU fallback)";
    case DABGPU_ERR_INVALID_ARG: return "invalid argument";
    case DABGPU_ERR_HIP: return "HIP runtime error";
    case DABGPU_ERR_NOT_READY: return "not ready";
    case DABGPU_ERR_UNSUPPORTED: return "unsupported transmission mode";
    default: return "unknown status";
    }
}
const char* dabgpu_last_error(void) { return g_last_error.c_str(); }
int dabgpu_abi_version(void) { return DABGPU_ABI_VERSION; }

// ---- built-in tables ----
// ETSI EN 300 401 14.3.2 tables 23/24 (mode I) and the mode II-IV tables of docs/DAB_implementation_in_SDR_detailed.pdf
// appendix B, as (row of the h table, offset n) per block of 32 carriers, lowest carrier first
// (replaces get_DAB_PRS_reference, src/ofdm/dab_prs_ref.cpp:25-195)
static const signed char PRS_ROW_I[4][48] = {
    { 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3, 0,1,2,3,  0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1, 0,3,2,1 },
    { 0,1,2,3,0,1,  2,1,0,3,2,1 },
    { 0,1,2,  3,2,1 },
    { 0,1,2,3, 0,1,2,3, 0,1,2,3,  0,3,2,1, 0,3,2,1, 0,3,2,1 },
};
static const signed char PRS_ROW_N[4][48] = {
    { 1,2,0,1, 3,2,2,3, 2,1,2,3, 1,2,3,3, 2,2,2,1, 1,3,1,2,  3,1,1,1, 2,2,1,0, 2,2,3,3, 0,2,1,3, 3,3,3,0, 3,0,1,1 },
    { 2,3,2,2,1,2,  0,2,2,1,0,3 },
    { 2,3,0,  2,2,2 },
    { 0,1,1,2, 2,2,0,3, 3,1,3,2,  0,1,0,2, 0,1,2,2, 2,1,3,0 },
};
static const signed char PRS_H_TABLE[4][32] = {
    {0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1,0,2,0,0,0,0,1,1,2,0,0,0,2,2,1,1},
    {0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0,0,3,2,3,0,1,3,0,2,1,2,3,2,3,3,0},
    {0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3,0,0,0,2,0,2,1,3,2,2,0,2,2,0,1,3},
    {0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2,0,1,2,1,0,3,3,2,2,3,2,1,2,1,3,2},
};

int dabgpu_get_prs_fft_ref(int mode, float* out) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    int geom[9];
    if (dabgpu_get_ofdm_params(mode, geom)) return DABGPU_ERR_INVALID_ARG;
    const int N = geom[3], nb = geom[5], rows = nb / 32, half = rows / 2;
    memset(out, 0, sizeof(float) * 2 * (size_t)N);
    for (int row = 0; row < rows; row++) {
        const int k_min = (row < half) ? (-nb / 2 + 32 * row) : (1 + 32 * (row - half));
        for (int j = 0; j < 32; j++) {
            const int k = k_min + j;
            const int h = PRS_H_TABLE[(int)PRS_ROW_I[mode - 1][row]][j];
            const float phi = (float)M_PI / 2.0f * (float)(h + PRS_ROW_N[mode - 1][row]);
            const int bin = (k < 0) ? (N + k) : k;
            out[2 * bin] = cosf(phi);
            out[2 * bin + 1] = sinf(phi);
        }
    }
    return DABGPU_OK;
}

// ETSI EN 300 401 14.6.1 (replaces get_DAB_mapper_ref, src/ofdm/dab_mapper_ref.cpp:10-51)
int dabgpu_get_carrier_mapper(int mode, int* out) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    int geom[9];
    if (dabgpu_get_ofdm_params(mode, geom)) return DABGPU_ERR_INVALID_ARG;
    const int N = geom[3], nb = geom[5], dc = N / 2, lo = dc - nb / 2, hi = dc + nb / 2;
    int v = 0, n = 0;
    for (int i = 0; i < N; i++) {
        if (i > 0) v = (13 * v + N / 4 - 1) % N;
        if (v < lo || v > hi || v == dc) continue;
        out[n++] = (v < dc) ? (v - lo) : (v - lo - 1);
    }
    return DABGPU_OK;
}

int dabgpu_get_fft_twiddles(float* out) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    for (int m = 0; m < DABGPU_NB_FFT; m++) {
        const double a = 2.0 * M_PI * (double)m / (double)DABGPU_NB_FFT;
        out[2 * m] = (float)cos(a);
        out[2 * m + 1] = (float)(-sin(a));
    }
    return DABGPU_OK;
}

void dabgpu_sync_cfg_default(dabgpu_sync_cfg* cfg) {          // ofdm_demodulator.h:34-44
    if (!cfg) return;
    cfg->fine_freq_update_beta = 0.9f;
    cfg->is_coarse_freq_correction = 1;
    cfg->max_coarse_freq_correction_norm = 0.5f;
    cfg->coarse_freq_slow_beta = 0.1f;
    cfg->impulse_peak_threshold_db = 20.0f;
    cfg->impulse_peak_distance_probability = 0.15f;
}

void dabgpu_stream_cfg_default(dabgpu_stream_cfg* c) {
    if (!c) return;
    c->signal_l1_update_beta = 0.95f; c->signal_l1_nb_samples = 100; c->signal_l1_nb_decimate = 5;      // ofdm_demodulator.h:25-29
    c->thresh_null_start = 0.35f; c->thresh_null_end = 0.75f;                                            // :30-33
    dabgpu_sync_cfg_default(&c->sync);
}

int dabgpu_get_ofdm_params(int mode, int* out9) {
    dabgpu::ModeGeom g;
    if (!out9 || !dabgpu::mode_geometry(mode, g)) { dabgpu_set_error("get_ofdm_params: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    out9[0] = g.n_sym; out9[1] = g.period; out9[2] = g.null_period; out9[3] = g.n_fft; out9[4] = g.n_cp; out9[5] = g.n_carriers;
    out9[6] = g.frame_samples; out9[7] = g.sym_bits; out9[8] = g.frame_bits;
    return DABGPU_OK;
}

int dabgpu_iq_format_from_mode(const char* mode) {
    static const char* const NAMES[14] = {"raw_u8", "raw_s8", "raw_s16l", "raw_s16b", "raw_u16l", "raw_u16b", "raw_s32l",
                                          "raw_s32b", "raw_u32l", "raw_u32b", "raw_f32l", "raw_f32b", "raw_f64l", "raw_f64b"};
    if (!mode) return -1;
    for (int i = 0; i < 14; i++) if (strcmp(mode, NAMES[i]) == 0) return i;
    return -1;
}

size_t dabgpu_iq_format_sample_bytes(int format) {
    if (format < 0 || format >= DABGPU_IQ_NB_FORMATS) return 0;
    // bytes per component: raw u8 s8 | s16l s16b u16l u16b | s32l s32b u32l u32b | f32l f32b | f64l f64b | wav pcm8 pcm16 pcm24 pcm32 f32 f64 alaw mulaw
    static const unsigned char SIZE[DABGPU_IQ_NB_FORMATS] = {1, 1, 2, 2, 2, 2, 4, 4, 4, 4, 4, 4, 8, 8, 1, 2, 3, 4, 4, 8, 1, 1};
    return 2 * (size_t)SIZE[format];
}

}  // extern "C"

// batches too small to fill the chip with three workgroups per frame: more, shorter runs -- a single frame in three runs of 25 symbols is
// three workgroups 25 symbols long (130 us); in 25 runs of 3 symbols (4 transforms each, one of them the halo) it is 25 workgroups 20 us
// long.  Aim at ~256 workgroups, at most 25 runs per frame; from 86 frames on the usual three runs.
int dabgpu_host_small_batch_spb(size_t n_frames) {
    if (n_frames >= 86) return 25;
    const size_t chunks = std::min<size_t>(25, (256 + n_frames - 1) / n_frames);
    return (int)((75 + chunks - 1) / chunks);
}

// size bucket of a batch: ceil(log2(n_frames)) -- 513..1024 frames share a bucket, 1025..2048 the next
int dabgpu_host_spb_bucket(size_t n_frames) {
    int b = 0;
    while (((size_t)1 << b) < n_frames && b < 40) b++;
    return b;
}
// kernel variant of a call: loader (0..3), soft-bit layout, whether the phase tail runs with it (fused at 75, a second launch otherwise)
int dabgpu_host_spb_variant(int src, int bits_layout, bool tail) { return src * 4 + (bits_layout == DABGPU_BITS_MSC_CLASSED ? 2 : 0) + (tail ? 1 : 0); }


// ETSI EN 300 401 tables 8 + 15: {size CU, kbps, level, L1..L4, PI1..PI4, padding bits}; row order (and the two
// exchanged size fields of rows 33/34) as the reference lists them, src/dab/constants/subchannel_protection_tables.h:21-86,
// because FIG 0/1 short-form sub-channels index this table by position
static const uint16_t UEP_ROWS[64][12] = {
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
// ETSI EN 300 401 tables 9/18 (EEP-A) and 10/20 (EEP-B): {CU multiple, m1, b1, m2, b2, PI1, PI2}, L = m*n + b;
// same data as subchannel_protection_tables.h:121-139
static const int EEP_A_ROWS[4][7] = { {12,6,-3,0,3,24,23}, {8,2,-3,4,3,14,13}, {6,6,-3,0,3,8,7}, {4,4,-3,2,3,3,2} };
static const int EEP_2A_N1[7] = { 8,0,5,0,1,13,12 };
static const int EEP_B_ROWS[4][7] = { {27,24,-3,0,3,10,9}, {21,24,-3,0,3,6,5}, {18,24,-3,0,3,4,3}, {15,24,-3,0,3,2,1} };

extern "C" int dabgpu_subchannel_plan(const dabgpu_subchannel* sc, int* pi, int* lx, int* n_decoded_bytes) {
    if (!sc || !pi || !lx) return -1;
    int nseg, total = 0;
    for (int i = 0; i < 4; i++) { pi[i] = 0; lx[i] = 0; }
    if (!sc->is_uep) {
        if (sc->eep_prot_level < 0 || sc->eep_prot_level > 3 || sc->length <= 0 || sc->length > 864) return -1;       // (a CIF has 864 capacity units)
        const int* d = (sc->eep_type == 0) ? ((sc->length == 8) ? EEP_2A_N1 : EEP_A_ROWS[sc->eep_prot_level])
                                            : EEP_B_ROWS[sc->eep_prot_level];       // GetEEPDescriptor :145-154
        const int n = sc->length / d[0];
        pi[0] = d[5]; lx[0] = d[1] * n + d[2];
        pi[1] = d[6]; lx[1] = d[3] * n + d[4];
        if (lx[0] < 0 || lx[1] < 0) return -1;
        nseg = 2;
    } else {
        if (sc->uep_prot_index < 0 || sc->uep_prot_index > 63) return -1;
        const uint16_t* d = UEP_ROWS[sc->uep_prot_index];
        for (int i = 0; i < 4; i++) { lx[i] = d[3 + i]; pi[i] = d[7 + i]; }
        nseg = 4;
    }
    for (int i = 0; i < nseg; i++) total += lx[i];
    if (n_decoded_bytes) *n_decoded_bytes = 4 * total;     // (32*sum(L) + 6 - 6) / 8, msc_decoder.cpp:99-103
    return nseg;
}


void dabgpu_host_fill_vit_tables(dabgpu_vit_tables* T) {
    memset(T, 0, sizeof(*T));
    static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    for (int pi = 1; pi <= 24; pi++) {
        int cnt[8];
        for (int g = 0; g < 8; g++) cnt[g] = 1;
        for (int e = 0; e < pi; e++) cnt[order[e % 8]]++;
        int pre = 0;
        for (int g = 0; g < 8; g++) { T->pi_tab[pi * 8 + g] = (uint16_t)(cnt[g] | (pre << 8)); pre += cnt[g]; }
    }
    unsigned reg = 0xFFFFu;
    for (int k = 0; k < 511; k++) {
        unsigned b = 0;
        for (int i = 0; i < 8; i++) {
            const unsigned v = ((reg >> 8) ^ (reg >> 4)) & 1u;
            b |= v << (7 - i);
            reg = ((reg << 1) | v) & 0xFFFFu;
        }
        T->prbs[k] = (unsigned char)b;
    }
}

int dabgpu_host_choose_mapping(int forced_mapping, double n_simd, size_t n_cw, size_t n_groups, double sum_cw_steps, double sum_group_steps, double max_steps,
                          bool staged_gather) {
    if (forced_mapping != DABGPU_VIT_MAP_AUTO) return forced_mapping;
    if (n_groups == 0) return DABGPU_VIT_MAP_WAVE;
    const double mean = sum_group_steps / (double)n_groups;
    const double gather = (staged_gather ? 3.3e-6 : 8.5e-6) * sum_cw_steps;
    // microseconds.  viterbi_kernel (re-fitted in round 5, after its chain-back went scalar): the latency of the longest code word on a lone wavefront
    // (~0.08 us per trellis step) + the throughput share of every code word -- 229 / 317 / 445 / 660 us for 32 / 64 / 100 / 160 ensembles of
    // 18 x 48 CU + FIC, 163 / 510 / 1926 us for the FIC of 1024 / 4096 / 16384 frames (tools/exp/map_crossover.py, profiles/r05/bench_fic_v4.json:
    // the fit is within 9 %).  The batch mappings: their rounds of wavefronts + the gather + ~20 us for their three set-up launches.
    const double t_wave = 0.08 * max_steps + 2.2e-5 * sum_cw_steps + 0.0113 * (double)n_cw;
    const double t_lane = 0.5 * std::max(max_steps, std::ceil((double)n_groups / n_simd) * mean) + gather + 20.0;
    const double t_oct = 0.095 * std::max(2.0 * max_steps, std::ceil(8.0 * (double)n_groups / n_simd) * mean) + gather + 20.0;
    if (t_wave <= t_lane && t_wave <= t_oct) return DABGPU_VIT_MAP_WAVE;
    return t_oct < t_lane ? DABGPU_VIT_MAP_OCTET : DABGPU_VIT_MAP_LANE;
}

// The MSC of n_ens ensembles that share a multiplex of n_sub sub-channels (steps[j] trellis steps each, 4 CIFs per call): the same cost
// model, groups = n_sub x ceil(4 n_ens / 64).  AUTO compares the three pure choices (a partial viterbi_kernel launch is one lockstep round of
// wavefronts and measured 2x its share of a full one: hybrids did not pay).  model_us (may be null) receives the modelled WAVE / LANE / OCTET
// times in microseconds.
int dabgpu_host_choose_msc_mapping(int forced_mapping, double n_simd, size_t n_ens, const uint32_t* steps, int n_sub, double* model_us) {
    double sum_steps = 0.0, max_st = 0.0;
    for (int j = 0; j < n_sub; j++) { sum_steps += (double)steps[j]; max_st = std::max(max_st, (double)steps[j]); }
    const double groups = (double)n_sub * (double)((n_ens * 4 + 63) / 64), mean = n_sub ? sum_steps / (double)n_sub : 0.0;
    const double gather = 3.3e-6 * sum_steps * (double)(n_ens * 4);
    const double t_wave = 0.08 * max_st + (double)(n_ens * 4) * (2.2e-5 * sum_steps + 0.0113 * (double)n_sub);
    const double t_lane = 0.5 * std::max(max_st, std::ceil(groups / n_simd) * mean) + gather + 20.0;
    const double t_oct = 0.095 * std::max(2.0 * max_st, std::ceil(8.0 * groups / n_simd) * mean) + gather + 20.0;
    if (model_us) { model_us[0] = t_wave; model_us[1] = t_lane; model_us[2] = t_oct; }
    if (forced_mapping != DABGPU_VIT_MAP_AUTO) return forced_mapping;
    if (t_lane < t_wave || t_oct < t_wave) return t_oct < t_lane ? DABGPU_VIT_MAP_OCTET : DABGPU_VIT_MAP_LANE;
    return DABGPU_VIT_MAP_WAVE;
}

int dabgpu_host_validate_codeword(const dabgpu_codeword& d, size_t i) {
    if (d.flags & DABGPU_CW_DEPUNCTURED) {            // mother code handed over: no segment tables, any length, direct source only
        if (d.n_steps < 1 || d.n_steps > DABGPU_MAX_TRELLIS_STEPS || d.n_slots != 0 || !d.d_src || !d.d_out) {
            dabgpu_set_error("codeword %zu: DABGPU_CW_DEPUNCTURED needs 1 <= n_steps <= %u, n_slots = 0 and non-null addresses", i, (unsigned)DABGPU_MAX_TRELLIS_STEPS);
            return DABGPU_ERR_INVALID_ARG;
        }
        return DABGPU_OK;
    }
    uint64_t steps = 0;
    for (int k = 0; k < 4; k++) {
        if (d.seg_steps[k] == 0) continue;
        if (d.seg_pi[k] < 1 || d.seg_pi[k] > 24 || (d.seg_steps[k] & 7)) {
            dabgpu_set_error("codeword %zu: segment %d has PI=%u steps=%u (PI must be 1..24, steps a multiple of 8)", i, k,
                             d.seg_pi[k], d.seg_steps[k]);
            return DABGPU_ERR_INVALID_ARG;
        }
        steps += d.seg_steps[k];
    }
    if (steps + 6 > DABGPU_MAX_TRELLIS_STEPS) {
        dabgpu_set_error("codeword %zu: %llu trellis steps, at most %u are decoded (the decoded bytes of a code word are assembled in LDS)", i,
                         (unsigned long long)steps + 6, (unsigned)DABGPU_MAX_TRELLIS_STEPS);
        return DABGPU_ERR_INVALID_ARG;
    }
    if (steps + 6 != d.n_steps || ((d.n_steps - 6) & 7) || !d.d_src || !d.d_out) {
        dabgpu_set_error("codeword %zu: n_steps=%u does not equal sum(seg_steps)+6 with whole output bytes, or null address", i, d.n_steps);
        return DABGPU_ERR_INVALID_ARG;
    }
    if ((d.flags & DABGPU_CW_CLASSED) && d.n_slots != 0 && (d.cif_stride == 0 || (d.cif_stride & 15))) {
        dabgpu_set_error("codeword %zu: DABGPU_CW_CLASSED needs cif_stride = soft bits per ring row, a multiple of 16 (got %u)", i, d.cif_stride);
        return DABGPU_ERR_INVALID_ARG;
    }
    if (d.n_slots != 0 && (d.n_slots < 16 || d.cifs_per_frame == 0 || d.newest_slot >= d.n_slots)) {
        dabgpu_set_error("codeword %zu: bad CIF ring geometry (n_slots=%u newest=%u cifs_per_frame=%u)", i, d.n_slots, d.newest_slot, d.cifs_per_frame);
        return DABGPU_ERR_INVALID_ARG;
    }
    return DABGPU_OK;
}


int dabgpu_host_build_msc_plans(const dabgpu_subchannel* h_sub, int n_sub, std::vector<dabgpu_msc_plan>& plans, uint32_t* cif_out_bytes,
                                uint32_t* max_steps_out, uint32_t* max_out_bytes) {
    if (!h_sub || n_sub < 0 || n_sub > 64) { dabgpu_set_error("msc_decode_frames: %d sub-channels (0..64 are accepted)", n_sub); return DABGPU_ERR_INVALID_ARG; }
    plans.assign((size_t)n_sub, dabgpu_msc_plan{});
    uint32_t off = 0, max_steps = 0, max_out = 0;
    for (int s = 0; s < n_sub; s++) {
        int pi[4], lx[4], nb = 0;
        // (ranges first: start + length of a hostile descriptor overflows int)
        if (h_sub[s].length <= 0 || h_sub[s].length > 864 || h_sub[s].start_address < 0 || h_sub[s].start_address > 864 - h_sub[s].length ||
            dabgpu_subchannel_plan(&h_sub[s], pi, lx, &nb) < 0) {
            dabgpu_set_error("msc_decode_frames: sub-channel %d has an invalid protection profile or exceeds 864 CU", s);
            return DABGPU_ERR_INVALID_ARG;
        }
        {   // a profile that consumes more soft bits than the sub-channel holds (a UEP table row paired with another size) would make the
            // gather read its neighbours: the reference's decoder runs out of symbols instead (dab_viterbi_decoder.cpp:157-160); refuse it
            uint32_t need = 12;
            for (int k = 0; k < 4; k++) need += 4u * (uint32_t)lx[k] * (8u + (uint32_t)pi[k]);
            if (need > (uint32_t)h_sub[s].length * 64u) {
                dabgpu_set_error("msc_decode_frames: sub-channel %d: its protection profile consumes %u soft bits, %d capacity units hold %u", s, need,
                                 h_sub[s].length, (uint32_t)h_sub[s].length * 64u);
                return DABGPU_ERR_INVALID_ARG;
            }
        }
        dabgpu_msc_plan& P = plans[(size_t)s];
        P.start_address = (uint32_t)h_sub[s].start_address;
        uint32_t steps = 0;
        for (int k = 0; k < 4; k++) { P.seg_pi[k] = lx[k] ? (uint32_t)pi[k] : 0u; P.seg_steps[k] = 32u * (uint32_t)lx[k]; steps += P.seg_steps[k]; }
        P.n_steps = steps + 6;
        P.out_offset = off;
        P.n_out_bytes = (uint32_t)nb;
        off += (uint32_t)nb;
        max_steps = std::max(max_steps, P.n_steps);
        max_out = std::max(max_out, (uint32_t)nb);
    }
    if (cif_out_bytes) *cif_out_bytes = off;
    if (max_steps_out) *max_steps_out = max_steps;
    if (max_out_bytes) *max_out_bytes = max_out;
    return DABGPU_OK;
}

extern "C" {
// ---- wav header (host only) ---------------------------------------------------------------------------------------
namespace {
struct byte_cursor {
    const uint8_t* p; size_t n; size_t pos;
    bool take(size_t k, const uint8_t** out) { if (n - pos < k) return false; *out = p + pos; pos += k; return true; }
};
inline uint32_t le32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
inline uint16_t le16(const uint8_t* b) { return (uint16_t)(b[0] | (b[1] << 8)); }
static int map_wav_code(uint16_t code, uint16_t* out) {
    switch (code) { case 1: case 3: case 6: case 7: case 0xFFFE: *out = code; return 1; default: return 0; }
}
}  // namespace

int dabgpu_wav_parse_header(const uint8_t* bytes, size_t n_bytes, dabgpu_wav_header* out) {
#define FAIL(...) do { dabgpu_set_error(__VA_ARGS__); return DABGPU_ERR_INVALID_ARG; } while (0)
    if (!bytes || !out) FAIL("wav_parse_header: null argument");
    byte_cursor cur{bytes, n_bytes, 0};
    const uint8_t* b;
    memset(out, 0, sizeof(*out));
    if (!cur.take(12, &b)) FAIL("wav: insufficient bytes while reading RIFF chunk");
    if (memcmp(b, "RIFF", 4) != 0) FAIL("wav: chunk id is not 'RIFF'");
    if (memcmp(b + 8, "WAVE", 4) != 0) FAIL("wav: wave id is not 'WAVE'");
    if (!cur.take(24, &b)) FAIL("wav: insufficient bytes while reading format chunk");
    if (memcmp(b, "fmt ", 4) != 0) FAIL("wav: chunk id is not 'fmt '");
    const uint32_t fmt_size = le32(b + 4);
    if (fmt_size != 16 && fmt_size != 18 && fmt_size != 40) FAIL("wav: invalid format chunk size %u, expected 16, 18 or 40", fmt_size);
    uint16_t code;
    if (!map_wav_code(le16(b + 8), &code)) FAIL("wav: invalid audio format code %04X", le16(b + 8));
    out->total_channels = le16(b + 10);
    if (out->total_channels != 1 && out->total_channels != 2) FAIL("wav: expected mono or stereo but got %u channels", out->total_channels);
    out->samples_per_second = le32(b + 12);
    out->average_bytes_per_second = le32(b + 16);
    out->data_block_align_bytes = le16(b + 20);
    out->bits_per_sample = le16(b + 22);
    if (fmt_size > 16) {
        const size_t ext = fmt_size - 16;
        if (!cur.take(ext, &b)) FAIL("wav: insufficient bytes while reading format chunk extension fields");
        const uint16_t ext_size = le16(b);
        if (ext_size != ext - 2) FAIL("wav: extension field size %u does not match actual size %zu", ext_size, ext - 2);
        if (ext_size == 22) {
            uint16_t sub;
            if (!map_wav_code(le16(b + 8), &sub)) FAIL("wav: invalid audio format code %04X", le16(b + 8));
            if (sub == 0xFFFE) FAIL("wav: extensible format again in sub-format");
            static const uint8_t GUID[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
            if (memcmp(GUID, b + 10, 14) != 0) FAIL("wav: extensible format guid does not match");
            code = sub;
        }
    }
    if (code != 1) {                                   // fact chunk for non-PCM formats
        if (!cur.take(8, &b)) FAIL("wav: insufficient bytes while reading fact chunk");
        if (memcmp(b, "fact", 4) != 0) FAIL("wav: chunk id is not 'fact'");
        const uint32_t fact_size = le32(b + 4);
        if (fact_size < 4) FAIL("wav: fact chunk smaller than 4 bytes (%u)", fact_size);
        if (!cur.take(fact_size, &b)) FAIL("wav: insufficient bytes while reading fact chunk data");
    }
    for (;;) {
        if (!cur.take(8, &b)) FAIL("wav: insufficient bytes while reading possible data chunk");
        const uint32_t size = le32(b + 4);
        if (memcmp(b, "data", 4) != 0) {
            // the reference fseek()s past the chunk and fails on the next header read when the file ends first
            if (cur.n - cur.pos < size) FAIL("wav: insufficient bytes while reading possible data chunk");
            cur.pos += size;
            continue;
        }
        out->data_chunk_size = size;
        out->data_chunk_offset = cur.pos;
        break;
    }
    out->audio_format = code;
    int f = -1;
    switch (code) {
    case 1:
        switch (out->bits_per_sample) {
        case 8: f = DABGPU_IQ_WAV_PCM8; break;
        case 16: f = DABGPU_IQ_WAV_PCM16; break;
        case 24: f = DABGPU_IQ_WAV_PCM24; break;
        case 32: f = DABGPU_IQ_WAV_PCM32; break;
        default: FAIL("wav: unhandled PCM format with %u bits per sample", out->bits_per_sample);
        }
        break;
    case 3:
        switch (out->bits_per_sample) {
        case 32: f = DABGPU_IQ_WAV_F32; break;
        case 64: f = DABGPU_IQ_WAV_F64; break;
        default: FAIL("wav: unhandled IEEE754 format with %u bits per sample", out->bits_per_sample);
        }
        break;
    case 6:
        if (out->bits_per_sample != 8) FAIL("wav: unhandled G711 A law format with %u bits per sample", out->bits_per_sample);
        f = DABGPU_IQ_WAV_ALAW; break;
    case 7:
        if (out->bits_per_sample != 8) FAIL("wav: unhandled G711 mu law format with %u bits per sample", out->bits_per_sample);
        f = DABGPU_IQ_WAV_MULAW; break;
    default: FAIL("wav: unhandled extensible wav audio format is not supported");
    }
    out->iq_format = f;
    return DABGPU_OK;
#undef FAIL
}

}  // extern "C"

extern "C" int dabgpu_subchannel_validate(const dabgpu_subchannel* sc) {
    if (!sc) { dabgpu_set_error("subchannel_validate: null descriptor"); return DABGPU_ERR_INVALID_ARG; }
    std::vector<dabgpu_msc_plan> one;
    return dabgpu_host_build_msc_plans(sc, 1, one, nullptr, nullptr, nullptr);
}
