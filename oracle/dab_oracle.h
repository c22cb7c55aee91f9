/*
 * dab_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, sequential restatement of the IQ-in / bits-out
 * hot path of williamyang98/DAB-Radio (reference snapshot 2025-08-29), used
 * ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under dab-radio_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Arithmetic contract:
 *   - IEEE-754 binary32, round-to-nearest-even, no implicit contraction
 *     (compile with -ffp-contract=off, no -ffast-math); every fused
 *     multiply-add is an explicit fmaf().
 *   - PLL / conj-mul element arithmetic = the reference's x86 AVX2+FMA build
 *     (its default preset, CMakePresets.json:77-78), pinned against objects
 *     compiled from the reference's own sources (oracle/_ref).
 *   - FFT: FFTW3 is an absent system dependency of the reference, so the
 *     2048-point transform is this repo's own Stockham 4x8x8x8 factorisation
 *     ("parity unpinned" at FFT rounding level; pinned on structure by DFT
 *     known answers and the TX->RX loop-back).
 *   - Viterbi ACS: vendor/viterbi_decoder (williamyang98/ViterbiDecoderCpp)
 *     is an empty submodule; restated from its published algorithm and the
 *     in-tree call sites ("parity unpinned" for tie-break / overflow corner
 *     cases; pinned by encode->decode round trips and FIB CRC16).
 */
#ifndef DAB_ORACLE_H
#define DAB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } dab_cf32;

/* ---- Mode I geometry (src/ofdm/dab_ofdm_params_ref.cpp:13-21) ---- */
enum {
    DAB_NB_FRAME_SYMBOLS = 76,
    DAB_NB_SYMBOL_PERIOD = 2552,
    DAB_NB_NULL_PERIOD   = 2656,
    DAB_NB_FFT           = 2048,
    DAB_NB_CYCLIC_PREFIX = 504,
    DAB_NB_DATA_CARRIERS = 1536,
    DAB_NB_FRAME_SAMPLES = 2656 + 76 * 2552,          /* 196608 */
    DAB_NB_SYM_BITS      = 3072,
    DAB_NB_FRAME_BITS    = 75 * 3072,                 /* 230400, dab_parameters.h:33 */
    DAB_NB_FIC_BITS      = 3 * 3072,                  /* 9216 */
    DAB_NB_FIB_GROUP_BITS= 2304,
    DAB_NB_CIF_BITS      = 55296,
    DAB_NB_CIFS          = 4
};

/* ---- constant tables ---- */
/* src/ofdm/dab_mapper_ref.cpp:10-51 : frequency de-interleaver, Mode I */
void dab_get_mapper(int *carrier_map /*[1536]*/);
/* src/ofdm/dab_prs_ref.cpp:140-195 : phase reference symbol spectrum, Mode I */
void dab_get_prs_fft(dab_cf32 *prs /*[2048]*/);
/* twiddle table used by the FFT contract: tw[m] = (cos, -sin)(2*pi*m/2048), rounded from double */
void dab_get_twiddles(dab_cf32 *tw /*[2048]*/);

/* ---- DSP primitives ---- */
/* src/ofdm/dsp/chebyshev_sine.h:22-41 scalar (mul+add) and :82-107 (AVX, __FMA__) */
float dab_chebyshev_sine(float x);      /* scalar, no FMA */
float dab_chebyshev_sine_fma(float x);  /* FMA Horner, as _mm256_chebyshev_sine with __FMA__ */
/* src/ofdm/dsp/apply_pll.cpp:12-30 (scalar reference build) */
void dab_apply_pll_scalar(const dab_cf32 *x, dab_cf32 *y, size_t n, float freq_norm, float dt_norm);
/* src/ofdm/dsp/apply_pll.cpp:81-117 + x86/c32_mul.h:9-38 (AVX + FMA reference build); THE contract */
void dab_apply_pll(const dab_cf32 *x, dab_cf32 *y, size_t n, float freq_norm, float dt_norm);
/* element of src/ofdm/dsp/x86/c32_conj_mul.h:12-44 with __FMA__: x0 * conj(x1) */
dab_cf32 dab_conj_mul(dab_cf32 x0, dab_cf32 x1);
/* src/ofdm/ofdm_demodulator.cpp:768-777 correlation part: sum_{n<504} sym[2048+n]*conj(sym[n]).
 * Element arithmetic as dab_conj_mul; summation order = this repo's fixed tree. */
dab_cf32 dab_cp_correlation(const dab_cf32 *sym /*[2552], PLL-corrected*/);
/* deterministic atan2 (Cephes-style), replaces std::atan2 at ofdm_demodulator.cpp:776 */
float dab_atan2f(float y, float x);
/* deterministic 20*log10(x) and 10^(x/20) used by the sync path (ofdm_demodulator.cpp:429,496,917) */
float dab_db20f(float magnitude);
float dab_undb20f(float db);
float dab_cabsf(dab_cf32 v);

/* forward / inverse unnormalised 2048-pt DFT, replaces FFTW (ofdm_demodulator.cpp:891-899) */
void dab_fft2048(const dab_cf32 *in, dab_cf32 *out);
void dab_ifft2048(const dab_cf32 *in, dab_cf32 *out);
/* O(N^2) double-precision DFT for known-answer checks of the above */
void dab_dft_naive(const dab_cf32 *in, double *out_re, double *out_im, int n, int inverse);

/* ---- OFDM demod of one frame-aligned frame (rows a8-a13 of SURVEY 8a) ---- */
/* src/ofdm/ofdm_demodulator.cpp:842-889 : DQPSK + de-interleave + soft-bit quantise, one symbol */
void dab_dqpsk_demap(const dab_cf32 *fft_i, const dab_cf32 *fft_ip1, const int *mapper, int8_t *bits /*[3072]*/);
/*
 * src/ofdm/ofdm_demodulator.cpp:650-766 with one pipeline thread: frame layout is the logical
 * OFDM_Frame_Buffer (ofdm_frame_buffer.h:87-99): 76 symbols x 2552 then the NULL symbol (2656).
 *   bits      [230400] out
 *   cp_corr   [76] out (optional): raw correlation sums per symbol
 *   cp_phase  [76] out (optional): atan2 of the above
 *   fft_out   [77*2048] out (optional): GetFrameFFT() content incl. NULL symbol
 * returns total phase error (sequential float sum over symbols 0..75, :685-691)
 */
float dab_demod_frame(const dab_cf32 *frame, float freq_offset, const int *mapper,
                      int8_t *bits, dab_cf32 *cp_corr, float *cp_phase, dab_cf32 *fft_out);
/* timing helper: n_total frame demods cycling over n_distinct frames (bits_scratch [230400]) */
void dab_demod_frames(const dab_cf32 *frames, size_t n_distinct, size_t n_total, float freq_offset, const int *mapper,
                      int8_t *bits_scratch, float *totals);
/* src/ofdm/ofdm_demodulator.cpp:606-618,779-840 : fine frequency IIR update */
float dab_update_fine_freq(float fine, float total_phase_error);
/* fmod wrap only (UpdateFineFrequencyOffset :829-840) */
float dab_fine_freq_add(float fine, float delta);

/* ---- sync (rows a5, a6) ---- */
typedef struct {
    /* OFDM_Demod_Config defaults ofdm_demodulator.h:24-45 */
    float fine_freq_update_beta;
    int   is_coarse_freq_correction;
    float max_coarse_freq_correction_norm;
    float coarse_freq_slow_beta;
    float impulse_peak_threshold_db;
    float impulse_peak_distance_probability;
} dab_sync_cfg;
void dab_sync_cfg_default(dab_sync_cfg *cfg);

typedef struct {
    float freq_coarse;
    float freq_fine;
    int   is_found_coarse;
    int   fine_time_offset;
    int   total_frames_read;
    int   total_frames_desync;
} dab_sync_state;

/* constructor-time references: ofdm_demodulator.cpp:128-140 */
void dab_sync_refs(const dab_cf32 *prs_fft, dab_cf32 *prs_fft_conj /*[2048]*/, dab_cf32 *prs_time_ref /*[2048]*/);
/* ofdm_demodulator.cpp:360-471 ; prs_sym = first 2048 samples of the PRS slot; freq_response[2048] optional */
void dab_coarse_freq_sync(const dab_cf32 *prs_sym, const dab_cf32 *prs_time_ref, const dab_sync_cfg *cfg,
                          dab_sync_state *st, float *freq_response);
/* ofdm_demodulator.cpp:473-548 ; returns 1 if peak valid, writes offset; impulse_response[2048] optional */
int dab_fine_time_sync(const dab_cf32 *prs_sym, const dab_cf32 *prs_fft_conj, const dab_sync_cfg *cfg,
                       float freq_offset, int *offset, float *impulse_response);

/* ---- channel decode ---- */
/* src/dab/constants/puncture_codes.h:42-72 */
const uint8_t *dab_puncture_code(int pi /*1..24*/);   /* 8 kept-counts */
const uint8_t *dab_puncture_code_tail(void);          /* PI_X, 6 kept-counts */
/* src/dab/algorithms/additive_scrambler.h:16-35 with syncword 0xFFFF */
void dab_scrambler_bytes(uint8_t *out, size_t n);
/* src/dab/algorithms/crc.h:25-68 specialised as fic_decoder.cpp:19-31 (poly 0x1021, init FFFF, xorout FFFF) */
uint16_t dab_crc16(const uint8_t *x, size_t n);

/* Viterbi decoder: restates DAB_Viterbi_Decoder (src/dab/algorithms/dab_viterbi_decoder.cpp:84-181)
 * over the published ViterbiDecoderCpp cores.  `tie_rule` is the CORE MODEL (the name is kept for the callers):
 *   0 = the scalar core (ViterbiDecoder_Scalar): uint16_t sums that WRAP, upper predecessor only if strictly smaller;
 *   1 = the SIMD cores (ViterbiDecoder_AVX_u16 / _SSE_u16 / _NEON_u16, what dab_viterbi_decoder.cpp:51-73 selects on an
 *       AVX2 / SSE4.1 / AArch64 build host): adds_epu16 sums that SATURATE at 65535, min_epu16 survivors, decision = cmpeq(survivor,
 *       upper candidate), i.e. the upper predecessor on ties.
 * Both remain restatements ("parity unpinned"): the vendor sources are an empty submodule here. */
typedef struct dab_viterbi dab_viterbi;
dab_viterbi *dab_viterbi_create(size_t traceback_length, int tie_rule);
void   dab_viterbi_destroy(dab_viterbi *v);
void   dab_viterbi_reset(dab_viterbi *v, size_t starting_state);
size_t dab_viterbi_update(dab_viterbi *v, const int8_t *punctured, size_t n_punctured,
                          const uint8_t *puncture_code, size_t n_code, size_t requested_output_symbols);
uint64_t dab_viterbi_chainback(dab_viterbi *v, uint8_t *bytes_out, size_t n_bytes, size_t end_state);
size_t dab_viterbi_current_decoded_bit(const dab_viterbi *v);
const uint64_t *dab_viterbi_decisions(const dab_viterbi *v);
const uint16_t *dab_viterbi_metrics(const dab_viterbi *v);

/* src/dab/fic/fic_decoder.cpp:53-117 : one FIB group (2304 soft bits) -> 96 bytes (3 x (30+2));
 * crc_ok_mask bit i set when FIB i passes CRC16. returns path error. */
uint64_t dab_fic_decode_group(const int8_t *bits /*[2304]*/, int tie_rule, uint8_t *bytes /*[96]*/, uint32_t *crc_ok_mask);

/* sub-channel descriptor: subset of Subchannel (src/dab/database/dab_database_entities.h:179-190) */
typedef struct {
    int start_address;   /* in CUs */
    int length;          /* in CUs */
    int is_uep;
    int uep_prot_index;
    int eep_prot_level;  /* 0..3 */
    int eep_type;        /* 0 = A, 1 = B */
} dab_subchannel;
/* segment plan per src/dab/msc/msc_decoder.cpp:77-154 + subchannel_protection_tables.h:
 * fills up to 4 (PI, Lx) pairs, returns count; n_decoded_bytes = output bytes per CIF */
int dab_subchannel_plan(const dab_subchannel *sc, int *pi /*[4]*/, int *lx /*[4]*/, int *n_decoded_bytes);
/* msc_decoder.cpp:77-154 on an already de-interleaved logical frame (length*64 soft bits) */
uint64_t dab_msc_decode_logical(const dab_subchannel *sc, const int8_t *bits, int tie_rule, uint8_t *bytes, int *n_bytes);

/* src/dab/msc/cif_deinterleaver.cpp:13-71 */
typedef struct dab_deinterleaver dab_deinterleaver;
dab_deinterleaver *dab_deinterleaver_create(int nb_bytes);
void dab_deinterleaver_destroy(dab_deinterleaver *d);
void dab_deinterleaver_consume(dab_deinterleaver *d, const int8_t *bits);
int  dab_deinterleaver_deinterleave(dab_deinterleaver *d, int8_t *out);

/* ---- transmit side: test-vector generator (no reference equivalent except the modulator) ---- */
/* ETSI EN 300 401 11.1: K=7 rate 1/4 mother code, polys 133,171,145,133 (dab_viterbi_decoder.cpp:15-25).
 * in: n_bits info bits (MSB-first in bytes); out: 4*(n_bits+6) mother code bits (0/1), 6 zero tail bits appended */
void dab_conv_encode(const uint8_t *bytes, size_t n_bits, uint8_t *mother /*[4*(n_bits+6)]*/);
/* inverse of depuncture_symbols (dab_viterbi_decoder.cpp:131-181) for one segment: returns #kept */
size_t dab_puncture(const uint8_t *mother, size_t n_mother, const uint8_t *code, size_t n_code, uint8_t *out);
/* FIB group encode: 3 x 30 data bytes -> CRC -> scramble -> conv -> puncture (PI_16 x21, PI_15 x3, PI_X) -> 2304 bits (0/1) */
void dab_fic_encode_group(const uint8_t *fib_data /*[90]*/, uint8_t *out_bits /*[2304]*/);
/* sub-channel logical frame encode (EEP/UEP): bytes -> scramble -> conv -> puncture -> length*64 bits */
void dab_msc_encode_logical(const dab_subchannel *sc, const uint8_t *bytes, uint8_t *out_bits);
/*
 * OFDM modulator: 230400 frame bits (0/1, layout SURVEY A.3) -> frequency interleave -> QPSK ->
 * differential -> IFFT -> CP -> NULL|PRS|75 symbols in *transmission order* (NULL first), amplitude as
 * src/ofdm/ofdm_modulator.cpp:49-156 (unnormalised IFFT of unit carriers).
 */
void dab_modulate_frame(const uint8_t *frame_bits /*[230400]*/, const int *mapper, dab_cf32 *out /*[196608]*/);
/* payload recipe of examples/simulate_transmitter.cpp:153-159 mapped by ofdm_modulator.cpp:95-156
 * (natural carrier order, 4 carriers per byte LSB dibit first) -> expected RX frame bits after
 * the receiver's de-interleave */
void dab_modulate_frame_reference_payload(const uint8_t *payload /*[28800]*/, dab_cf32 *out /*[196608]*/);

/* ---------------------------------------------------------------------------------------------
 * Data formats either side of the path.
 * IQ readers: examples/app_helpers/app_iq_readers.h:17-159 + app_wav_reader.h:257-456; format numbers follow
 * the order of iq_read_modes (app_iq_readers.h:107-113) after "wav", then the wav encodings:
 *  0 raw_u8 1 raw_s8 2 raw_s16l 3 raw_s16b 4 raw_u16l 5 raw_u16b 6 raw_s32l 7 raw_s32b 8 raw_u32l 9 raw_u32b
 *  10 raw_f32l 11 raw_f32b 12 raw_f64l 13 raw_f64b 14 wav pcm8 15 pcm16 16 pcm24 17 pcm32 18 f32 19 f64 20 A-law 21 mu-law
 */
#define DAB_IQ_NB_FORMATS 22
size_t dab_iq_component_bytes(int format);
/* n_comp components (2 per IQ sample) -> floats; returns 0, or -1 for an unknown format */
int dab_iq_convert(const uint8_t *raw, int format, size_t n_comp, float *out);
/* app_viterbi_convert_block.h:12-44 */
void dab_hard_bytes_to_soft_bits(const uint8_t *bytes, size_t n_bytes, int8_t *bits);
void dab_soft_bits_to_hard_bytes(const int8_t *bits, size_t n_bytes, uint8_t *bytes);
/* wav_read_header (app_wav_reader.h:107-255) + the WavFileReader constructor's format checks (:268-466) over a
 * file image; out7 = {format number above, audio format code, channels, samples_per_second, bits_per_sample,
 * data_chunk_size, data_chunk_offset}; returns 0, or -1 where the reference throws */
int dab_wav_parse_header(const uint8_t *bytes, size_t n_bytes, uint64_t *out7);

/* ---------------------------------------------------------------------------------------------
 * DAB+ outer code: audio super-frame assembly, fire code, RS(120,110), access-unit CRCs
 * (src/dab/audio/aac_frame_processor.cpp, src/dab/algorithms/reed_solomon_decoder.cpp; ETSI TS 102 563 clauses 5, 6)
 */
/* RS(255,245) over GF(2^8), p(x) = x^8+x^4+x^3+x^2+1, roots alpha^0..alpha^9, shortened by 135 leading zero symbols.
 * Corrects cw[120] in place; returns the number of located errors (0..5) or -1; positions[] (may be NULL) receives the
 * located symbol indices in the 255-symbol padded block (index - 135 = byte of cw), in Chien-search order. An error
 * located inside the padding is counted but not applied (reed_solomon_decoder.cpp:474-478). */
int dab_rs120_decode(uint8_t *cw, int *positions);
void dab_rs120_encode(const uint8_t *data110, uint8_t *parity10);          /* test helper */
uint16_t dab_firecode_crc(const uint8_t *data9);                           /* aac_frame_processor.cpp:75-86 */

typedef struct {
    int32_t superframe_done;        /* this call completed a super frame attempt (5th logical frame) */
    int32_t firecode_wait_failed;   /* WAIT_FRAME_START: the frame's fire code did not match, frame dropped (:162-166) */
    int32_t rs_failed_index;        /* -1, or the index of the RS codeword that was uncorrectable (:336-341) */
    int32_t rs_corrected;           /* total symbols reported corrected (sum of Decode() returns) */
    int32_t firecode_ok;            /* fire code of the corrected super frame (:206-209); 0 when RS failed */
    int32_t header_valid;           /* header parsed, access units walked */
    int32_t descriptor;             /* byte 2 of the super frame */
    int32_t num_aus;
    int32_t au_start[8];            /* au_start[0..num_aus] */
    int32_t au_walk_stopped_at;     /* -1, or the AU index at which the bounds check aborted the walk (:291-297) */
    uint32_t au_crc_ok_mask;        /* bit i: access unit i passed its CRC (:304-314) */
} dab_superframe_result;

typedef struct dab_aac_frame_processor dab_aac_frame_processor;
dab_aac_frame_processor *dab_aac_create(void);
void dab_aac_destroy(dab_aac_frame_processor *p);
/* AAC_Frame_Processor::Process (:127-176); sf_out receives the (corrected) super frame bytes [5*n] when
 * superframe_done; returns 0, or -1 when the reference rejects the buffer (empty / shorter than 11 bytes) */
int dab_aac_process(dab_aac_frame_processor *p, const uint8_t *frame, int n, dab_superframe_result *res, uint8_t *sf_out);

/* ---------------------------------------------------------------------------------------------
 * Transmission modes II-IV
 */
typedef struct {
    int mode, nb_frame_symbols, nb_symbol_period, nb_null_period, nb_fft, nb_cp, nb_carriers;
    int nb_frame_samples, nb_sym_bits, nb_frame_bits;
} dab_ofdm_geometry;
int dab_ofdm_geometry_get(int mode, dab_ofdm_geometry *g);
void dab_mapper_n(int nb_fft, int nb_carriers, int *out);
void dab_fft_n(int n, const dab_cf32 *in, dab_cf32 *out, int inverse);      /* n in {256, 512, 1024, 2048} */
dab_cf32 dab_cp_correlation_n(const dab_cf32 *sym, int nb_fft, int nb_cp);
float dab_demod_frame_mode(int mode, const dab_cf32 *frame, float f, const int *mapper, int8_t *bits, dab_cf32 *cp_corr,
                           float *cp_phase, dab_cf32 *fft_out);
float dab_update_fine_freq_mode(int mode, float fine, float total_phase_error, float beta);
/* get_DAB_PRS_reference (src/ofdm/dab_prs_ref.cpp:140-195) and the PRS synchronisation of OFDM_Demod for any mode; the mode I
 * instances equal dab_get_prs_fft / dab_sync_refs / dab_coarse_freq_sync / dab_fine_time_sync bit for bit */
int dab_get_prs_fft_mode(int mode, dab_cf32 *prs /*[nb_fft]*/);
void dab_sync_refs_mode(int mode, const dab_cf32 *prs_fft, dab_cf32 *prs_fft_conj, dab_cf32 *prs_time_ref);
void dab_coarse_freq_sync_mode(int mode, const dab_cf32 *prs_sym, const dab_cf32 *prs_time_ref, const dab_sync_cfg *cfg,
                               dab_sync_state *st, float *freq_response);
int dab_fine_time_sync_mode(int mode, const dab_cf32 *prs_sym, const dab_cf32 *prs_fft_conj, const dab_sync_cfg *cfg,
                            float freq_offset, int *offset, float *impulse_response);

/* ---------------------------------------------------------------------------------------------
 * One receiver's per-frame sequence as one call (dab_oracle_chain.c): n_total frames cycling over n_distinct stored slices of `stride`
 * samples whose PRS is expected at sample prs_offset (>= 504; stride >= prs_offset + 1544 + 196608): coarse + fine synchronisation,
 * demodulation at the position found with the tracked offset, fine-frequency update, the 4 FIB groups, and every listed sub-channel of
 * the 4 CIFs through its own CIF_Deinterleaver and MSC decode.  `state` persists across calls (zero it for a new receiver).
 * fib_last [4][96] / msc_last [4][sum of decoded bytes per CIF] (optional) receive the last frame's outputs; returns 0, or -1 on
 * invalid arguments.  bench.py's cpu_baseline_full times this on every host core. */
int dab_receive_frames(const dab_cf32 *slices, size_t n_distinct, size_t stride, size_t prs_offset, size_t n_total,
                       const dab_subchannel *subs, int n_subs, int tie_rule, dab_sync_state *state,
                       uint32_t *n_fib_crc_ok, uint32_t *n_sync_failed, uint8_t *fib_last, uint8_t *msc_last, uint64_t *digest);

#ifdef __cplusplus
}
#endif
#endif
