/*
 * dabgpu.h -- C ABI of the MI355X-native DAB OFDM-demodulation + channel-decode hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, `int` status
 * returns (0 = ok, see dabgpu_strerror), no exceptions cross it.  Each entry point names the
 * reference interface (williamyang98/DAB-Radio @ 2025-08-29, paths relative to the repo root)
 * it replaces.  Host-side C++ classes with the reference's own names and signatures
 * (OFDM_Demod, FIC_Decoder, MSC_Decoder, DAB_Viterbi_Decoder, CIF_Deinterleaver) are layered on
 * top of this ABI in dab-radio_amd/host/ ; INTEGRATION.md shows how basic_radio links them.
 *
 * Pointers prefixed d_ are DEVICE pointers (hipMalloc / torch tensor data_ptr), h_ are host.
 * `stream` is a hipStream_t passed as void* (NULL = the HIP default stream, as in HIP itself).  All launches
 * are asynchronous on that stream unless the function name ends in _sync (those use a private stream of the
 * context and return when the results are in host memory).
 *
 * HIP graphs: the batch entry points (dabgpu_ofdm_demod_frames*, dabgpu_ofdm_sync_demod_frames, dabgpu_fic_decode_frames,
 * dabgpu_msc_decode_frames*, dabgpu_decode_frames_layout) enqueue kernels only once they have run ONCE with the same shapes and the
 * same sub-channel list on the stream (scratch is grow-only, tables that depend on the sub-channel list are uploaded when they change):
 * such calls may be issued between hipStreamBeginCapture and hipStreamEndCapture.  A call that would have to upload a table during a
 * capture returns DABGPU_ERR_INVALID_ARG instead of invalidating it, and so does one that would have to grow the context's device scratch.
 * A captured graph holds the addresses of the context's scratch and tables as they were at capture time.  It stays VALID while the context
 * lives -- a later eager call with larger shapes gets new scratch, the outgrown buffers are kept until dabgpu_destroy once a call of the
 * context has run under capture -- and it stays CORRECT only while no call with another sub-channel list, another mapping
 * (dabgpu_viterbi_set_mapping, DABGPU_VIT_HYBRID_K) or larger shapes runs on the same context between its replays: those rewrite the plan and
 * lane tables the graph's kernels read.  One context per captured pipeline.  (tests/test_gpu_graph_capture.py)
 *
 * All functions fail with DABGPU_ERR_NO_DEVICE when no gfx950 device is usable: there is no CPU
 * fallback behind this ABI.
 */
#ifndef DABGPU_H
#define DABGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: symbols_per_block = 0 no longer measures (dabgpu_ofdm_tune does, explicitly); receiver pipeline (dabgpu_receiver_*)
 * 3: dabgpu_receiver_submit_demod / _submit_decode (the OFDM_Demod mirror class needs them) */
#define DABGPU_ABI_VERSION 4

/* Mode I geometry (src/ofdm/dab_ofdm_params_ref.cpp:13-21, src/dab/constants/dab_parameters.h:31-40) */
#define DABGPU_NB_FRAME_SYMBOLS 76
#define DABGPU_NB_SYMBOL_PERIOD 2552
#define DABGPU_NB_NULL_PERIOD   2656
#define DABGPU_NB_FFT           2048
#define DABGPU_NB_CYCLIC_PREFIX 504
#define DABGPU_NB_DATA_CARRIERS 1536
#define DABGPU_NB_FRAME_SAMPLES 196608
#define DABGPU_NB_SYM_BITS      3072
#define DABGPU_NB_FRAME_BITS    230400
#define DABGPU_NB_FIC_BITS      9216
#define DABGPU_NB_FIB_GROUP_BITS 2304
#define DABGPU_NB_CIF_BITS      55296
#define DABGPU_NB_CIFS          4

enum {
    DABGPU_OK = 0,
    DABGPU_ERR_NO_DEVICE = 1,      /* no HIP device / not gfx950 / runtime missing */
    DABGPU_ERR_INVALID_ARG = 2,
    DABGPU_ERR_HIP = 3,            /* a HIP call failed; text via dabgpu_last_error() */
    DABGPU_ERR_NOT_READY = 4,      /* e.g. time de-interleaver has fewer than 16 CIFs */
    DABGPU_ERR_UNSUPPORTED = 5     /* transmission mode other than I */
};
enum { DABGPU_CORE_SCALAR = 0, DABGPU_CORE_SIMD = 1 };    /* values of the `tie_rule` argument: "CORE MODEL" at the channel decoder below */

typedef struct dabgpu_ctx dabgpu_ctx;

const char *dabgpu_strerror(int status);
const char *dabgpu_last_error(void);          /* thread-local detail of the last failure */
int dabgpu_abi_version(void);
/* number of usable gfx950 devices (0 when none; never initialises a HIP context on failure) */
int dabgpu_device_count(void);

/*
 * Context = one device + its constant tables (FFT twiddles, inverse carrier map, conj(PRS), coarse-sync
 * time reference).  Replaces the constructor-time work of OFDM_Demod::OFDM_Demod
 * (src/ofdm/ofdm_demodulator.cpp:80-146): h_prs_fft_ref is get_DAB_PRS_reference() output
 * (2048 complex float, interleaved re/im), h_carrier_mapper is get_DAB_mapper_ref() output (1536 int).
 * Passing NULL for either uses the built-in Mode I tables.
 */
int dabgpu_create(dabgpu_ctx **out, int device, const float *h_prs_fft_ref, const int *h_carrier_mapper);
void dabgpu_destroy(dabgpu_ctx *ctx);
/* Page-lock / release a host buffer that is handed to the *_host_sync entry points repeatedly (hipHostRegister): the copies then run at
 * PCIe speed.  Optional -- unpinned buffers work, slower. */
int dabgpu_host_pin(void *h_buffer, size_t bytes);
int dabgpu_host_unpin(void *h_buffer);
int dabgpu_synchronize(dabgpu_ctx *ctx, void *stream);

/* Built-in Mode I tables, host side (replace get_DAB_PRS_reference src/ofdm/dab_prs_ref.cpp:140,
 * get_DAB_mapper_ref src/ofdm/dab_mapper_ref.cpp:10, and expose the FFT twiddle table of the arithmetic contract) */
int dabgpu_get_prs_fft_ref(int transmission_mode, float *h_out /*[2*2048]*/);
int dabgpu_get_carrier_mapper(int transmission_mode, int *h_out /*[1536]*/);
int dabgpu_get_fft_twiddles(float *h_out /*[2*2048]*/);

/* ------------------------------------------------------------------------------------------------
 * OFDM demodulation of frame-aligned frames: PLL + cyclic-prefix phase error + 2048-pt FFT + DQPSK +
 * frequency de-interleave + soft-bit quantisation, batched over frames.
 * Replaces OFDM_Demod::PipelineThread (src/ofdm/ofdm_demodulator.cpp:650-766) and its callees
 * ApplyPLL, CalculateCyclicPhaseError, CalculateFFT, CalculateDQPSK, CalculateViterbiBits.
 *
 *   d_iq          [n_frames][196608] complex float; frame layout = OFDM_Frame_Buffer's logical layout
 *                 (src/ofdm/ofdm_frame_buffer.h:87-99): 76 symbols x 2552 samples (PRS first) then the
 *                 NULL symbol (2656).  16-byte aligned.
 *   d_freq_offset [n_frames] net normalised frequency offset used by the PLL (coarse + fine), may be NULL (= 0)
 *   d_bits        [n_frames][230400] int8 soft bits, layout of On_OFDM_Frame() (ofdm_demodulator.cpp:635)
 *   d_cp_corr     [n_frames][76] complex float raw cyclic-prefix correlation per symbol, may be NULL
 *   d_fft         [n_frames][77][2048] complex float = GetFrameFFT() content, may be NULL (skips the
 *                 display-only NULL-symbol FFT, ofdm_demodulator.cpp:701-709)
 *   d_dqpsk       [n_frames][75][1536] complex float = GetFrameDataVec() content (X_i * conj(X_{i+1}) per carrier,
 *                 natural carrier order, ofdm_demodulator.cpp:842-865), may be NULL
 *   symbols_per_block  data symbols handled by one workgroup, 1..75; any value gives identical results.  0 = the library chooses, WITHOUT
 *                 measuring (the call stays asynchronous): batches below 512 frames take 25, below 86 frames shorter runs (down to 3) so
 *                 that the batch still spreads over the chip; larger ones take what dabgpu_ofdm_tune recorded for this kernel variant in
 *                 the nearest size bucket, else 25.  Which of 25 / 38 / 75 is fastest depends on the box.
 *   bits_frame_stride  bytes between the soft bits of consecutive frames (0 = 230400, packed); lets the kernel write
 *                 straight into slot k of a per-ensemble frame-history ring (see dabgpu_msc_decode_frames)
 */
int dabgpu_ofdm_demod_frames(dabgpu_ctx *ctx, const float *d_iq, size_t n_frames, const float *d_freq_offset,
                             int8_t *d_bits, float *d_cp_corr, float *d_fft, float *d_dqpsk, int symbols_per_block,
                             size_t bits_frame_stride, void *stream);

/*
 * The same with the frames still in their capture format (dabgpu_iq_format below; n_frames x 196608 IQ samples, 16-byte
 * aligned).  For raw_u8 / raw_s8 / raw_s16l (and wav PCM8 / PCM16 / float32 payloads) the kernel's loader reads the raw
 * samples straight from HBM and dequantises them in registers with the reader arithmetic of
 * examples/app_helpers/app_iq_readers.h:19-44,79-84 -- 2 or 4 bytes per sample instead of 8 and no conversion pass;
 * every other format is converted into context scratch on `stream` first.  Results are bit-identical to
 * dabgpu_iq_convert followed by dabgpu_ofdm_demod_frames.
 */
int dabgpu_ofdm_demod_frames_raw(dabgpu_ctx *ctx, const void *d_raw, int format, size_t n_frames, const float *d_freq_offset,
                                 int8_t *d_bits, float *d_cp_corr, float *d_fft, float *d_dqpsk, int symbols_per_block,
                                 size_t bits_frame_stride, void *stream);

/*
 * Demodulation straight into the frame-history ring the channel decoder reads (dabgpu_msc_decode_frames), with a choice of
 * how the MSC soft bits are laid out there.  Formats with a fused loader only (float32, u8, s8, s16 little endian, and the wav
 * payloads of those); soft bits and the cyclic-prefix correlation only.
 *   DABGPU_BITS_NATURAL      the layout of On_OFDM_Frame() (ofdm_demodulator.cpp:635): identical to dabgpu_ofdm_demod_frames_raw
 *   DABGPU_BITS_MSC_CLASSED  FIC (soft bits 0..9215) as above; inside each of the four CIF rows of 55296 soft bits, bit i is
 *                            stored at (i mod 16) * 3456 + i / 16.  CIF_Deinterleaver takes the bits of class i mod 16 from the
 *                            CIF that is 15 - bitrev4(i mod 16) CIFs old (src/dab/msc/cif_deinterleaver.cpp:57-68); in this order
 *                            each of those reads is contiguous, which cuts the history traffic of the decoder's gather from
 *                            19/4 of the decoded bits to 1 (the permutation itself costs nothing: it rides on the frequency
 *                            de-interleave the kernel performs anyway).  Only the *_layout decoders below read it.
 */
#define DABGPU_BITS_NATURAL 0
#define DABGPU_BITS_MSC_CLASSED 1
int dabgpu_ofdm_demod_frames_history(dabgpu_ctx *ctx, const void *d_raw, int format, size_t n_frames, const float *d_freq_offset,
                                     int8_t *d_bits, float *d_cp_corr, int symbols_per_block, size_t bits_frame_stride,
                                     int bits_layout, void *stream);

/*
 * dabgpu_ofdm_demod_frames_history followed by dabgpu_ofdm_phase_update(d_cp_corr, ...) as one call -- and as ONE launch whenever a
 * workgroup walks a whole frame (symbols_per_block = 75: explicit, or the library's own choice for this batch size): the phase tail
 * (per-symbol atan2, sequential sum, fine-frequency update: ofdm_demodulator.cpp:606-618, :779-840) then runs at the end of the
 * demodulation kernel, on the correlations it has just produced.  Otherwise the tail follows as its own launch.  Results are
 * identical to the two separate calls.  d_total_phase / d_fine_freq as in dabgpu_ofdm_phase_update (either may be NULL).
 */
int dabgpu_ofdm_demod_phase_frames(dabgpu_ctx *ctx, const void *d_raw, int format, size_t n_frames, const float *d_freq_offset,
                                   int8_t *d_bits, float *d_cp_corr, int symbols_per_block, size_t bits_frame_stride, int bits_layout,
                                   float fine_freq_update_beta, float *d_total_phase, float *d_fine_freq, void *stream);

/*
 * Calibration of symbols_per_block = 0, explicit and out of the data path: times the run lengths 25 / 38 / 75 of the demodulation of
 * n_frames frames (>= 512; smaller batches follow a fixed rule and return at once) on the caller's buffers -- ~40 ms of warm-up launches,
 * then three timed rounds over the candidates -- and records the fastest for (loader of `format`, bits_layout, with_phase_tail, size bucket
 * = ceil(log2(n_frames))).  BLOCKS the calling thread (hipEventSynchronize on `stream`; refused while the stream is capturing); d_bits
 * receives the frames' soft bits for a zero carrier offset; with_phase_tail times the candidates with the phase tail of dabgpu_ofdm_demod_phase_frames (on context
 * scratch, not on caller state).  *chosen (may be NULL) = the run length recorded.  Typical use: once at start-up per batch size.
 */
int dabgpu_ofdm_tune(dabgpu_ctx *ctx, const void *d_raw, int format, size_t n_frames, int8_t *d_bits, size_t bits_frame_stride,
                     int bits_layout, int with_phase_tail, void *stream, int *chosen);
/* what symbols_per_block = 0 resolves to for that call shape right now (never 0 for valid arguments) */
int dabgpu_ofdm_tuned_symbols_per_block(dabgpu_ctx *ctx, int format, size_t n_frames, int bits_layout, int with_phase_tail);
/* the run length most recently recorded by dabgpu_ofdm_tune for the size bucket of n_frames, whatever the variant: 25 / 38 / 75
 * (3 .. 25 below 86 frames: the fixed rule), or 0 = nothing recorded for that bucket */
int dabgpu_ofdm_auto_symbols_per_block(dabgpu_ctx *ctx, size_t n_frames);

/*
 * Per-frame scalar tail of the fine-frequency loop: phase[i] = atan2(corr[i]), total = sum_i phase[i]
 * (sequential, i = 0..75), and optionally fine <- fmod(fine - beta*err, wrap).
 * Replaces OFDM_Demod::CoordinatorThread's phase section (ofdm_demodulator.cpp:606-618) +
 * CalculateFineFrequencyError (:779-824) + UpdateFineFrequencyOffset (:829-840).
 *   d_cp_corr     [n_frames][76] complex float (from dabgpu_ofdm_demod_frames)
 *   d_total_phase [n_frames] out, may be NULL
 *   d_fine_freq   [n_frames] in/out, may be NULL (no update)
 */
int dabgpu_ofdm_phase_update(dabgpu_ctx *ctx, const float *d_cp_corr, size_t n_frames, float fine_freq_update_beta,
                             float *d_total_phase, float *d_fine_freq, void *stream);

/* Host-buffer convenience: H2D of the frames, demod, D2H of bits (+ totals); synchronous.
 * h_total_phase may be NULL. This is what the single-stream OFDM_Demod mirror class calls per frame. */
int dabgpu_ofdm_demod_frames_host_sync(dabgpu_ctx *ctx, const float *h_iq, size_t n_frames, const float *h_freq_offset,
                                       int8_t *h_bits, float *h_total_phase, float *h_fft);

/*
 * One frame of one stream, host buffers, synchronous, including the fine-frequency loop update: the PLL runs with
 * freq_coarse + *h_freq_fine, then *h_freq_fine <- fmod(*h_freq_fine - beta * err, wrap) on the device
 * (src/ofdm/ofdm_demodulator.cpp:606-618, :829-840).  h_fft [77][2048] and h_dqpsk [75][1536] complex float may be NULL.
 * This is the per-frame call of the OFDM_Demod mirror class (replaces CoordinatorThread + PipelineThread for one frame).
 */
int dabgpu_ofdm_demod_stream_frame_sync(dabgpu_ctx *ctx, const float *h_iq, float freq_coarse, float *h_freq_fine,
                                        float fine_freq_update_beta, int8_t *h_bits, float *h_total_phase, float *h_fft,
                                        float *h_dqpsk);

/* ------------------------------------------------------------------------------------------------
 * PRS synchronisation: integer-bin frequency offset (coarse) then symbol timing from the impulse response (fine).
 * Replaces OFDM_Demod::RunCoarseFreqSync and ::RunFineTimeSync (src/ofdm/ofdm_demodulator.cpp:360-471, :473-548),
 * one workgroup per stream.  `dabgpu_sync_cfg` mirrors OFDM_Demod_Config::sync (src/ofdm/ofdm_demodulator.h:34-44).
 */
typedef struct {
    float fine_freq_update_beta;              /* 0.9  */
    int   is_coarse_freq_correction;          /* 1    */
    float max_coarse_freq_correction_norm;    /* 0.5  */
    float coarse_freq_slow_beta;              /* 0.1  */
    float impulse_peak_threshold_db;          /* 20   */
    float impulse_peak_distance_probability;  /* 0.15 */
} dabgpu_sync_cfg;

typedef struct {
    float freq_coarse;        /* in/out: m_freq_coarse_offset */
    float freq_fine;          /* in/out: m_freq_fine_offset */
    int   is_found_coarse;    /* in/out: m_is_found_coarse_freq_offset */
    int   fine_time_offset;   /* out: m_fine_time_offset (PRS start relative to the expected position), when sync_valid */
    int   sync_valid;         /* out: 0 = impulse peak below threshold -> the caller must Reset() (:529-532) */
    int   reserved;
} dabgpu_sync_state;

void dabgpu_sync_cfg_default(dabgpu_sync_cfg *cfg);
/*
 *   d_prs_syms   stream s: 2048 complex float starting at d_prs_syms + s * stride_samples (complex samples) =
 *                m_correlation_time_buffer[nb_null_period ...], i.e. the first 2048 samples of the expected PRS slot
 *   d_states     [n_streams] in/out
 *   d_impulse_response / d_freq_response  [n_streams][2048] float dB, may be NULL
 *                (GetImpulseResponse() / GetCoarseFrequencyResponse())
 */
int dabgpu_ofdm_sync(dabgpu_ctx *ctx, const float *d_prs_syms, size_t n_streams, size_t stride_samples,
                     const dabgpu_sync_cfg *cfg, dabgpu_sync_state *d_states, float *d_impulse_response,
                     float *d_freq_response, void *stream);
/* single stream from host memory, synchronous; used by the OFDM_Demod mirror class once per frame */
int dabgpu_ofdm_sync_host_sync(dabgpu_ctx *ctx, const float *h_prs_sym, const dabgpu_sync_cfg *cfg,
                               dabgpu_sync_state *h_state, float *h_impulse_response, float *h_freq_response);

/*
 * One steady-state frame of n receivers in ONE call: PRS synchronisation at the expected position, demodulation from where the impulse
 * peak puts the frame with the offset the synchroniser has just tracked, fine-frequency update -- OFDM_Demod's per-frame sequence
 * RunCoarseFreqSync -> RunFineTimeSync -> PipelineThread (PLL with m_freq_coarse_offset + m_freq_fine_offset) -> CoordinatorThread's
 * UpdateFineFrequencyOffset (src/ofdm/ofdm_demodulator.cpp:360-471, :473-548, :650-766, :606-618), the sync records staying on the device:
 * dabgpu_ofdm_sync followed by a demodulation that reads d_states[k] for its position, its PLL offset and its fine-frequency word.
 *   d_iq      receiver k's samples: complex float at d_iq + 2 * k * stream_stride_samples (8-byte aligned); sample prs_offset_samples of the
 *             slice is where the receiver expects the PRS to begin (what m_correlation_time_buffer[nb_null_period] holds).  The frame's 76
 *             symbols are read from prs_offset_samples + fine_time_offset on; fine_time_offset lies in [-504, 1543], hence
 *             prs_offset_samples >= 504 and stream_stride_samples >= prs_offset_samples + 1544 + 76 * 2552 (also for the last receiver).
 *   d_states  [n_streams] in/out: freq_coarse / freq_fine / is_found_coarse persist from frame to frame; after the call freq_fine holds
 *             the value updated by this frame's cyclic-prefix phase error.  sync_valid = 0: the impulse-peak test failed, nothing was
 *             demodulated for that receiver (d_bits / d_cp_corr / d_total_phase rows untouched) and the caller resets it (:529-532).
 *   d_bits, d_cp_corr (may be NULL), symbols_per_block, bits_frame_stride, bits_layout: as dabgpu_ofdm_demod_frames_history
 *   d_total_phase [n_streams] sum of the 76 cyclic-prefix angles, may be NULL;  beta of the update = cfg->fine_freq_update_beta
 * Results equal dabgpu_ofdm_sync + dabgpu_ofdm_demod_frames_history on the shifted frames + dabgpu_ofdm_phase_update, bit for bit.
 */
int dabgpu_ofdm_sync_demod_frames(dabgpu_ctx *ctx, const float *d_iq, size_t n_streams, size_t stream_stride_samples,
                                  size_t prs_offset_samples, const dabgpu_sync_cfg *cfg, dabgpu_sync_state *d_states, int8_t *d_bits,
                                  float *d_cp_corr, int symbols_per_block, size_t bits_frame_stride, int bits_layout,
                                  float *d_total_phase, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Channel decoding: punctured K=7 rate-1/4 Viterbi (+ time de-interleave, energy dispersal, FIB CRC16),
 * one wavefront per codeword, batched.
 * Replaces DAB_Viterbi_Decoder::update/chainback (src/dab/algorithms/dab_viterbi_decoder.cpp:114-181) and the
 * vendor/viterbi_decoder core behind it, AdditiveScrambler (src/dab/algorithms/additive_scrambler.h:16-35),
 * CRC_Calculator<uint16_t> as used by the FIC (src/dab/fic/fic_decoder.cpp:19-31,103-116) and
 * CIF_Deinterleaver::Deinterleave (src/dab/msc/cif_deinterleaver.cpp:36-71).
 *
 * tie_rule = the CORE MODEL (the argument keeps its round-1 name): which of the upstream ViterbiDecoderCpp cores the decoder restates.
 *   DABGPU_CORE_SCALAR (0)  ViterbiDecoder_Scalar: uint16_t candidate sums that WRAP, the upper predecessor only when strictly smaller;
 *   DABGPU_CORE_SIMD   (1)  ViterbiDecoder_AVX_u16 / _SSE_u16 / _NEON_u16 -- what src/dab/algorithms/dab_viterbi_decoder.cpp:51-73 selects on
 *                           an AVX2 / SSE4.1 / AArch64 build host, i.e. the reference's default -march=native build on x86: adds_epu16 sums
 *                           that SATURATE at 65535, min_epu16 survivors, decision = cmpeq(survivor, upper candidate): the upper predecessor on ties.
 * For this code and soft bits in [-127, 127] no candidate sum can reach 65535 (at most 60454 + 254 * 19 = 65280: tests/test_independent_pins.py::
 * test_u16_candidate_sums_cannot_reach_65535 derives the bound from the polynomials and attacks it), so the two models give different bytes only
 * where two candidates tie exactly; both are implemented in full anyway.  Neither is pinned to upstream output here (vendor/viterbi_decoder is an
 * empty submodule): "parity unpinned" for the core, DESIGN.md 3.6.  The C++ classes pick the model the reference's build would have picked on
 * the host they run on (DABGPU_VITERBI_CORE=scalar|simd overrides; DABGPU_TIE_RULE=0|1 is the older spelling).
 * Soft bits are int8 in [-127,+127], 0 = punctured/erased (src/viterbi_config.h:11-14); -128 is read as -127.
 */
typedef struct {
    uint64_t d_src;            /* device address. direct (n_slots == 0): first soft bit of the codeword;
                                  ring: first soft bit of this sub-channel inside CIF slot 0 */
    uint64_t d_out;            /* device address of the decoded, descrambled bytes ((n_steps-6)/8 of them) */
    uint32_t n_steps;          /* trellis steps = information bits + 6 tail bits */
    uint32_t seg_pi[4];        /* puncturing vector index PI (1..24) of up to 4 segments, 0 = unused
                                  (src/dab/constants/puncture_codes.h:42-72); the 24-symbol tail PI_X is implicit */
    uint32_t seg_steps[4];     /* trellis steps of each segment = 32 * L (L = number of 128-bit blocks) */
    uint32_t start_state;      /* DAB_Viterbi_Decoder::reset(starting_state) */
    uint32_t n_crc_blocks;     /* > 0: output is that many equal blocks each ending in a CRC16 (FIBs) */
    uint32_t n_slots;          /* 0 = direct; else length of the CIF ring in slots (>= 16) */
    uint32_t newest_slot;      /* ring slot holding the most recent CIF of this codeword */
    uint32_t cifs_per_frame;   /* ring geometry: slot s lives at d_src + (s / cifs_per_frame) * frame_stride */
    uint32_t frame_stride;     /*                                      + (s % cifs_per_frame) * cif_stride   (bytes) */
    uint32_t cif_stride;
    uint32_t end_state;        /* DAB_Viterbi_Decoder::chainback(bytes_out, end_state), normally 0 */
    uint32_t flags;            /* DABGPU_CW_RAW: emit the decoder output without the energy-dispersal XOR */
} dabgpu_codeword;
/* longest code word any decoder entry point takes, in trellis steps (= information bits + 6): the one-wavefront-per-code-word kernel
 * assembles a code word's decoded bytes in LDS (60 KB).  17x the longest DAB code word (a sub-channel of all 864 capacity units at rate 8/9:
 * 27,648 steps); longer inputs return DABGPU_ERR_INVALID_ARG. */
#define DABGPU_MAX_TRELLIS_STEPS 491526u
#define DABGPU_CW_RAW 1u
#define DABGPU_CW_DEPUNCTURED 8u  /* direct codewords only (n_slots == 0): d_src holds the MOTHER code, 4 soft bits per trellis step with the punctured
                                  positions already 0 (the caller ran DAB_Viterbi_Decoder::depuncture_symbols, dab_viterbi_decoder.cpp:131-181,
                                  for whatever puncture vectors and lengths it was given); seg_pi / seg_steps are ignored, n_steps may be any
                                  value in 1 .. DABGPU_MAX_TRELLIS_STEPS, (n_steps - 6) / 8 whole bytes are written.  Always decoded by the WAVE mapping. */
#define DABGPU_CW_CLASSED 4u   /* ring codewords only: every ring row holds its cif_stride soft bits in time-interleaver class order;
                                  input bit i of a slot lives at d_src + slot offset + (i mod 16) * (cif_stride / 16) + i / 16, and
                                  d_src points at the sub-channel's first byte of class 0 (slot 0) */

typedef struct {
    uint64_t path_error;       /* DAB_Viterbi_Decoder::chainback() return value (dab_viterbi_decoder.cpp:124-129) */
    uint32_t crc_ok_mask;      /* bit i = CRC16 of block i matches (only when n_crc_blocks > 0) */
    uint32_t n_out_bytes;
} dabgpu_codeword_result;

/*
 * Three device mappings of the same decoder (identical results, bit for bit):
 *   WAVE   one wavefront per codeword, the 64 trellis states across its lanes (viterbi.hip) -- any batch, any mix of schedules
 *   LANE   one lane per codeword, wavefronts of 64 codewords that share a puncturing schedule (viterbi_lanes.hip) -- ~3x the
 *          throughput once a batch holds thousands of codewords per schedule; needs <= 768 bytes of scratch per trellis step
 *          and 64 codewords
 *   OCTET  eight lanes per codeword, 8 codewords per wavefront (viterbi_octet.hip), over the same groups and scratch as LANE --
 *          for the batches in between: when LANE would leave most SIMDs without a wavefront (64 codewords are indivisible there)
 * AUTO (default) compares a cost model of the three for the call at hand (LANE needs ~0.5 us per trellis step of its longest
 * schedule however small the batch, OCTET ~0.19 us, WAVE ~0.08 us per step of the longest codeword + 0.022 ns per codeword and step +
 * 0.011 us per codeword): e.g. the FIC goes WAVE -> OCTET at ~1000 frames and OCTET -> LANE at ~10000, the 18 x 48 CU multiplex
 * WAVE -> OCTET at ~65 ensembles and OCTET -> LANE at ~560
 * (dabgpu_viterbi_decode_batch: only when every codeword of the batch has the same n_steps / segments; forcing LANE or OCTET on
 * a mixed generic batch runs WAVE).
 */
#define DABGPU_VIT_MAP_AUTO 0
#define DABGPU_VIT_MAP_WAVE 1
#define DABGPU_VIT_MAP_LANE 2
#define DABGPU_VIT_MAP_OCTET 3
int dabgpu_viterbi_set_mapping(dabgpu_ctx *ctx, int mapping);

/* generic batch: h_codewords is a HOST array (copied to the device on `stream`); d_results a DEVICE array [n] */
int dabgpu_viterbi_decode_batch(dabgpu_ctx *ctx, const dabgpu_codeword *h_codewords, size_t n, int tie_rule,
                                dabgpu_codeword_result *d_results, void *stream);

/*
 * FIC of whole frames: for every frame the 4 FIB groups of 2304 soft bits (first 9216 bits of the frame) are
 * decoded with PI_16 x21, PI_15 x3, PI_X, descrambled and CRC-checked.
 * Replaces BasicFICRunner::Process + FIC_Decoder::DecodeFIBGroup (src/basic_radio/basic_fic_runner.cpp:34-49,
 * src/dab/fic/fic_decoder.cpp:53-117) for a batch.
 *   d_bits       frame f starts at d_bits + f * frame_stride (bytes); pass DABGPU_NB_FRAME_BITS for packed frames
 *   d_fib_bytes  [n_frames][4][96]: 3 x (30 data bytes + 2 CRC bytes) per group
 *   d_results    [n_frames][4]
 * dabgpu_fic_decode_* keep their device scratch apart from dabgpu_msc_decode_* / dabgpu_viterbi_decode_batch: one context may
 * decode the FIC and the MSC of a batch concurrently on two streams (the FIC fits into the SIMD time the MSC leaves idle).
 */
int dabgpu_fic_decode_frames(dabgpu_ctx *ctx, const int8_t *d_bits, size_t n_frames, size_t frame_stride,
                             uint8_t *d_fib_bytes, dabgpu_codeword_result *d_results, int tie_rule, void *stream);

/* Sub-channel description: the fields of `Subchannel` the decoder reads (src/dab/database/dab_database_entities.h:179-190) */
typedef struct {
    int start_address;   /* capacity units */
    int length;          /* capacity units */
    int is_uep;
    int uep_prot_index;  /* 0..63, row of the UEP table */
    int eep_prot_level;  /* 0..3 = level 1..4 */
    int eep_type;        /* 0 = EEP-A, 1 = EEP-B */
} dabgpu_subchannel;

/* (PI, L) plan of a sub-channel as MSC_Decoder::DecodeEEP/DecodeUEP derive it (src/dab/msc/msc_decoder.cpp:77-131,
 * src/dab/constants/subchannel_protection_tables.h); returns the number of segments (<= 4) or -1 */
int dabgpu_subchannel_plan(const dabgpu_subchannel *sc, int *pi4, int *l4, int *n_decoded_bytes);
/* DABGPU_OK when the decoders accept the descriptor: a profile of the tables, inside the 864 capacity units of a CIF, and consuming no more
 * soft bits than the sub-channel holds (UEP table row 34 as the reference lists it does: the reference's decoder runs out of symbols inside
 * its third update(), dab_viterbi_decoder.cpp:157-160; here the descriptor is refused); else DABGPU_ERR_INVALID_ARG with the reason in
 * dabgpu_last_error().  Host only, no device needed: what MSC_Decoder's constructor checks before anything is created (ABI 4). */
int dabgpu_subchannel_validate(const dabgpu_subchannel *sc);

/*
 * MSC of whole frames for many ensembles that share one multiplex configuration.  The time de-interleaver reads
 * straight from the history of demodulated frames: d_bits_history holds, per ensemble, `history_frames` (>= 5)
 * frame slots of 230400 soft bits used as a ring; the frame most recently written is slot `newest_frame_slot`.
 * Every call decodes the 4 CIFs of that newest frame for every listed sub-channel (the first 3 frames after a
 * (re)start yield garbage until 16 CIFs are in the ring -- the caller tracks that, as CIF_Deinterleaver's
 * m_total_frames_stored does, cif_deinterleaver.cpp:28-42).
 * Replaces MSC_Decoder::DecodeCIF (src/dab/msc/msc_decoder.cpp:46-115) x 4 CIFs x sub-channels x ensembles.
 *   ensemble e's history starts at d_bits_history + e * ensemble_stride (bytes)
 *   output of (ensemble e, cif c, sub-channel s) is written at d_out + e * out_ensemble_stride + c * B + off_s where
 *   B = sum of decoded bytes of all listed sub-channels and off_s the running sum; d_results is [n_ensembles][4][n_sub]
 */
int dabgpu_msc_decode_frames(dabgpu_ctx *ctx, const int8_t *d_bits_history, size_t n_ensembles, size_t ensemble_stride,
                             int history_frames, int newest_frame_slot, const dabgpu_subchannel *h_subchannels,
                             int n_subchannels, uint8_t *d_out, size_t out_ensemble_stride,
                             dabgpu_codeword_result *d_results, int tie_rule, void *stream);

/* The same reading a history whose MSC part was written in `bits_layout` (dabgpu_ofdm_demod_frames_history) */
int dabgpu_msc_decode_frames_layout(dabgpu_ctx *ctx, const int8_t *d_bits_history, size_t n_ensembles, size_t ensemble_stride,
                                    int history_frames, int newest_frame_slot, const dabgpu_subchannel *h_subchannels,
                                    int n_subchannels, uint8_t *d_out, size_t out_ensemble_stride,
                                    dabgpu_codeword_result *d_results, int tie_rule, int bits_layout, void *stream);

/*
 * FIC + MSC of one transmission frame of every ensemble in ONE call: dabgpu_fic_decode_frames on ring slot newest_frame_slot and
 * dabgpu_msc_decode_frames_layout on the ring, with the arguments and outputs of those two (identical bytes and result records).
 * Replaces BasicRadio::Process's fan-out of a frame to its FIC runner and its MSC runners (src/basic_radio/basic_radio.cpp:41-65).
 * When every sub-channel of the batch runs in a batch mapping (LANE / OCTET) the FIB groups are decoded INSIDE the MSC launch, as
 * further groups of codewords with their own puncturing schedule -- the MSC's groups rarely fill their last round of wavefront
 * slots, so the FIC then costs its gather only; otherwise the two run one after the other as if called separately.
 *   d_fib_bytes [n_ensembles][4][96], d_fic_results [n_ensembles][4]; d_msc_out / d_msc_results as in dabgpu_msc_decode_frames.
 */
int dabgpu_decode_frames_layout(dabgpu_ctx *ctx, const int8_t *d_bits_history, size_t n_ensembles, size_t ensemble_stride,
                                int history_frames, int newest_frame_slot, const dabgpu_subchannel *h_subchannels, int n_subchannels,
                                uint8_t *d_fib_bytes, dabgpu_codeword_result *d_fic_results, uint8_t *d_msc_out,
                                size_t out_ensemble_stride, dabgpu_codeword_result *d_msc_results, int tie_rule, int bits_layout,
                                void *stream);

/* Which mapping the MSC of `n_ensembles` ensembles carrying this multiplex takes right now -- the context's setting (dabgpu_viterbi_set_mapping) or,
 * under DABGPU_VIT_MAP_AUTO, the cost model's choice (one mapping for all sub-channels of a call) -- and the modelled times of WAVE / LANE / OCTET in
 * microseconds (model_us3 may be NULL).  Host only: launches nothing. */
int dabgpu_multiplex_mapping(dabgpu_ctx *ctx, size_t n_ensembles, const dabgpu_subchannel *h_subchannels, int n_subchannels, int *mapping,
                             double *model_us3);

/* Ring forms: ensemble e decodes the frame in slot d_newest_slot[e] of its own frame-history ring d_hist + e*ensemble_stride
 * (each ensemble at its own ring position, as dabgpu_stream_bank_process_ring leaves them); a negative slot skips the ensemble
 * (its result records come back with n_out_bytes = 0).  FIB bytes [n_ensembles][4][96], results [n_ensembles][4]. */
int dabgpu_fic_decode_ring(dabgpu_ctx *ctx, const int8_t *d_hist, size_t n_ensembles, size_t ensemble_stride,
                           const int32_t *d_newest_slot, uint8_t *d_fib_bytes, dabgpu_codeword_result *d_results, int tie_rule,
                           void *stream);
int dabgpu_msc_decode_ring(dabgpu_ctx *ctx, const int8_t *d_hist, size_t n_ensembles, size_t ensemble_stride, int history_frames,
                           const int32_t *d_newest_slot, const dabgpu_subchannel *subchannels, int n_subchannels, uint8_t *d_out,
                           size_t out_ensemble_stride, dabgpu_codeword_result *d_results, int tie_rule, void *stream);

/* dabgpu_decode_frames_layout for rings (dabgpu_fic_decode_ring + dabgpu_msc_decode_ring_layout in one call) */
int dabgpu_decode_ring_layout(dabgpu_ctx *ctx, const int8_t *d_hist, size_t n_ensembles, size_t ensemble_stride, int history_frames,
                              const int32_t *d_newest_slot, const dabgpu_subchannel *subchannels, int n_subchannels,
                              uint8_t *d_fib_bytes, dabgpu_codeword_result *d_fic_results, uint8_t *d_msc_out, size_t out_ensemble_stride,
                              dabgpu_codeword_result *d_msc_results, int tie_rule, int bits_layout, void *stream);

/* msc_decode_ring for rings whose MSC part is in `bits_layout` (dabgpu_stream_bank_process_ring_layout) */
int dabgpu_msc_decode_ring_layout(dabgpu_ctx *ctx, const int8_t *d_hist, size_t n_ensembles, size_t ensemble_stride, int history_frames,
                                  const int32_t *d_newest_slot, const dabgpu_subchannel *subchannels, int n_subchannels, uint8_t *d_out,
                                  size_t out_ensemble_stride, dabgpu_codeword_result *d_results, int tie_rule, int bits_layout,
                                  void *stream);

/* ------------------------------------------------------------------------------------------------
 * Single-stream, host-buffer, synchronous forms used by the C++ mirror classes (one codeword per call).
 */
/* one FIB group: FIC_Decoder::DecodeFIBGroup (src/dab/fic/fic_decoder.cpp:53-117) */
int dabgpu_fic_decode_group_host_sync(dabgpu_ctx *ctx, const int8_t *h_bits /*[2304]*/, uint8_t *h_bytes /*[96]*/,
                                      uint32_t *crc_ok_mask, uint64_t *path_error, int tie_rule);
/* DAB_Viterbi_Decoder reset/update.../chainback collapsed into one call: h_src holds the punctured soft bits of all
 * segments back to back (src/dab/algorithms/dab_viterbi_decoder.cpp:109-129) */
int dabgpu_viterbi_decode_host_sync(dabgpu_ctx *ctx, const int8_t *h_src, size_t n_src, const uint32_t *seg_pi4,
                                    const uint32_t *seg_steps4, uint32_t start_state, uint32_t end_state, uint32_t flags,
                                    uint8_t *h_out, size_t n_out_bytes, uint64_t *path_error, int tie_rule);

/* The general form of DAB_Viterbi_Decoder -- reset(start_state), update(...) any number of times with ANY puncture vector and ANY
 * requested_output_symbols, chainback(bytes_out, end_state) (src/dab/algorithms/dab_viterbi_decoder.cpp:109-181) -- collapsed into one call:
 * the caller de-punctures as it goes (kept symbols copied, punctured positions 0: :154-176) and hands over the mother code.
 *   h_mother     [4 * n_steps] int8, 1 <= n_steps <= DABGPU_MAX_TRELLIS_STEPS = trellis steps since reset()
 *   n_out_bytes  bytes chainback() is asked for: its bits n_out_bytes * 8 - 1 .. 0 are traced back from decision word n_out_bytes * 8 + 5
 *                downwards, starting in end_state (the core's chainback behind :126); n_out_bytes * 8 + 6 <= n_steps (a trace-back that
 *                starts beyond the decoded steps reads decision words the reference never wrote in this run: DABGPU_ERR_INVALID_ARG)
 *   path_error   accumulated renormalisation + metric[end_state] after ALL n_steps steps (:127-128), may be NULL.  (The reference calls the
 *                core's get_error() WITHOUT an argument, dab_viterbi_decoder.cpp:127; which state that reads by default is defined in
 *                vendor/viterbi_decoder, an empty submodule here.  Every in-tree caller chains back from end_state 0, where metric[end_state]
 *                and metric[0] are the same number; for other end states the two readings may differ -- unpinned, like the oracle's.)
 * When the trace-back does not start at the last decoded step the forward pass runs twice (whole length for the path error, the prefix
 * for the bytes): the decisions of a prefix do not depend on what follows. */
int dabgpu_viterbi_decode_depunctured_host_sync(dabgpu_ctx *ctx, const int8_t *h_mother, size_t n_steps, uint32_t start_state,
                                                uint32_t end_state, uint8_t *h_out, size_t n_out_bytes, uint64_t *path_error, int tie_rule);

/* per-sub-channel stream: a 16-slot device ring of the sub-channel's slice of every CIF + its protection plan.
 * Replaces MSC_Decoder + CIF_Deinterleaver state (src/dab/msc/msc_decoder.cpp:26-75, cif_deinterleaver.cpp:13-34). */
typedef struct dabgpu_msc_stream dabgpu_msc_stream;
int dabgpu_msc_stream_create(dabgpu_ctx *ctx, const dabgpu_subchannel *sc, dabgpu_msc_stream **out);
void dabgpu_msc_stream_destroy(dabgpu_msc_stream *s);
/* CIF_Deinterleaver::Consume: h_bits = length*64 soft bits of this sub-channel from one CIF */
int dabgpu_msc_stream_push_cif(dabgpu_msc_stream *s, const int8_t *h_bits);
/* CIF_Deinterleaver::Deinterleave: DABGPU_ERR_NOT_READY until 16 CIFs were pushed */
int dabgpu_msc_stream_deinterleave_sync(dabgpu_msc_stream *s, int8_t *h_out);
/* Deinterleave + DecodeEEP/DecodeUEP of the logical frame completed by the last push; *n_out = decoded bytes.
 * DABGPU_ERR_NOT_READY (and *n_out = 0) until 16 CIFs were pushed (msc_decoder.cpp:60-63) */
int dabgpu_msc_stream_decode_sync(dabgpu_msc_stream *s, uint8_t *h_out, size_t *n_out, uint64_t *path_error, int tie_rule);

/* --------------------------------------------------------------------------------------------------
 * Frame session: one batched decode per transmission frame behind the single-stream classes.
 * BasicRadio::Process fans a frame out to one FIC runner and one MSC runner per sub-channel (src/basic_radio/basic_radio.cpp:41-65),
 * each of which calls its decoder once per FIB group / CIF -- 4 + 4 x sub-channels synchronous round trips per frame.  A session keeps
 * the last 8 frames of soft bits on the device: push_frame copies a frame in ONCE and launches the FIC decode (4 FIB groups) and the
 * time de-interleave + Viterbi + descramble of every registered sub-channel for the frame's 4 CIFs (the batch entry points above with
 * one ensemble), asynchronously; the results of the last 8 frames stay available in host memory and are fetched by
 * (generation, group / sub-channel, CIF).  The mirror classes use it through dab-radio_amd/host/dab/dabgpu_frame_batcher.h, which
 * also decides when a class may take a session result instead of decoding itself (the bytes are identical either way).
 * Thread-safe.  The session owns a device context of its own.
 */
typedef struct dabgpu_frame_session dabgpu_frame_session;
int dabgpu_frame_session_create(dabgpu_frame_session **out, int device);
void dabgpu_frame_session_destroy(dabgpu_frame_session *s);
/* replaces the set of sub-channels decoded for the frames pushed from now on (n <= 64) */
int dabgpu_frame_session_set_subchannels(dabgpu_frame_session *s, const dabgpu_subchannel *h_subchannels, int n);
/* h_bits = the 230400 soft bits of one frame (layout of On_OFDM_Frame()); *generation counts the frames pushed, from 0 */
int dabgpu_frame_session_push_frame(dabgpu_frame_session *s, const int8_t *h_bits, int decode_fic, int tie_rule, uint64_t *generation);
/* DABGPU_ERR_NOT_READY: that generation is gone (more than 8 frames old), was pushed without the FIC / without this sub-channel */
int dabgpu_frame_session_fetch_fib_group(dabgpu_frame_session *s, uint64_t generation, int group, uint8_t *h_bytes /*[96]*/,
                                         uint32_t *crc_ok_mask, uint64_t *path_error);
int dabgpu_frame_session_fetch_cif(dabgpu_frame_session *s, uint64_t generation, const dabgpu_subchannel *sc, int cif,
                                   uint8_t *h_bytes, size_t capacity, size_t *n_bytes, uint64_t *path_error);

/* --------------------------------------------------------------------------------------------------
 * Receiver pipeline: one receiver's per-frame device work without a host wait in between.
 * Replaces the hand-over between OFDM_Demod's reader thread, its coordinator / pipeline threads and the observers that decode the frame
 * (src/ofdm/ofdm_demodulator.cpp:550-577 double buffer + WaitEnd / SignalStart, :581-639 CoordinatorThread; src/basic_radio/basic_radio.cpp:41-65).
 * The OFDM_Demod mirror class is built on it (dab-radio_amd/host/ofdm/ofdm_demodulator.cpp): its reader side buffers samples into a
 * pinned staging buffer and submits, a delivery thread waits for frames and calls the observers.
 *
 *   staging buffer   dabgpu_receiver_stage(): page-locked host memory the caller assembles "NULL symbol | frame" in, as complex float;
 *                    capacity = nb_null_period + (nb_fft - nb_cyclic_prefix) + samples per frame.  dabgpu_receiver_submit_frame moves on to
 *                    the next of three buffers; the one just submitted stays readable (the caller copies the frame's trailing NULL symbol
 *                    out of it for the next correlation window, ofdm_demodulator.cpp:558-562).
 *   frequency state  m_freq_coarse_offset / m_freq_fine_offset / m_is_found_coarse_freq_offset live on the DEVICE: submit_sync runs
 *                    RunCoarseFreqSync + RunFineTimeSync (:360-548) on them, submit_frame demodulates with their sum (:672) and applies the
 *                    frame's fine-frequency update (:606-618, :829-840), dabgpu_receiver_reset zeroes them (:277-289) -- all in submission
 *                    order on one stream, so frame k + 1's synchroniser sees frame k's update without the host having seen it.
 *   decode           frames of a mode I receiver are demodulated straight into the 8-frame history of a frame session
 *                    (dabgpu_receiver_session; see "Frame session" above) and decoded there on a second stream -- the FIC when
 *                    decode_fic, the sub-channels given -- the results fetched with dabgpu_frame_session_fetch_* by generation.
 * Threads: submit_* / stage / reset / set_subchannels / wait_sync from ONE thread (the reader); wait_frame from one other (or the same).
 * At most 7 frames may be submitted and not yet collected with wait_frame (the result slots are a ring of 8).
 */
/*
 * BANKED receivers (ABI 4): dabgpu_receiver_create_banked makes a mode I receiver on the library's own tables that is a member of the device's
 * RECEIVER BANK.  Its interface is the one below, unchanged; its submit_* calls post the work, and one thread per device issues what all members
 * have posted since its last round as ONE synchroniser launch, ONE demodulation launch over a compact batch of the posted frames and ONE FIC + MSC
 * decode over the members' history rings (at most one job per member and round, in posting order: frame k's fine-frequency update still
 * precedes frame k + 1's synchroniser).  A member uploads its frame itself when it posts it (the samples cross PCIe while earlier rounds run); at most
 * two rounds are under way (DABGPU_BANK_ROUNDS), what is posted meanwhile forms the next one -- its size follows the load.  N OFDM_Demod objects of a
 * process then cost ~4 runtime calls per frame and member + ~15 per round instead of ~20 per frame on three streams each (csrc/receiver_bank.hip;
 * DESIGN.md 4.11b).  Outputs are those of the private pipeline, bit for bit.
 * The sub-channel list (dabgpu_receiver_set_subchannels) is the BANK's: the members report one list (the decoders' subscription is process-wide in
 * the classes above); it applies to the rounds enqueued after the call.  A round decodes what it demodulates: frames are posted with dabgpu_receiver_submit_frame (which carries the
 * core model); the two-call form dabgpu_receiver_submit_demod / _submit_decode returns DABGPU_ERR_UNSUPPORTED.  Up to 64 members per device.
 */
typedef struct dabgpu_receiver dabgpu_receiver;
typedef struct {
    uint64_t generation;
    const int8_t *bits;        /* the frame's soft bits, On_OFDM_Frame() layout, in page-locked memory of the receiver: valid until 7 more frames were submitted */
    size_t n_bits;
    float freq_fine;           /* m_freq_fine_offset after this frame's update */
    float total_phase;         /* sum of the frame's cyclic-prefix angles */
    const float *fft;          /* GetFrameFFT() [nb_frame_symbols + 1][nb_fft] complex, only when the frame was submitted with want_views */
    const float *dqpsk;        /* GetFrameDataVec() [nb_frame_symbols - 1][nb_data_carriers] complex, mode I + want_views */
} dabgpu_receiver_frame;
/* h_prs_fft_ref / h_carrier_mapper as for dabgpu_create (mode I; NULL = built-in tables; must be NULL in modes II-IV) */
int dabgpu_receiver_create(dabgpu_receiver **out, int device, int transmission_mode, const float *h_prs_fft_ref, const int *h_carrier_mapper);
int dabgpu_receiver_create_banked(dabgpu_receiver **out, int device);
void dabgpu_receiver_destroy(dabgpu_receiver *rx);
dabgpu_frame_session *dabgpu_receiver_session(dabgpu_receiver *rx);      /* owned by the receiver */
/* what is decoded for the frames submitted from now on (mode I; n <= 64) */
int dabgpu_receiver_set_subchannels(dabgpu_receiver *rx, const dabgpu_subchannel *h_subchannels, int n, int decode_fic);
int dabgpu_receiver_stage(dabgpu_receiver *rx, float **h_stage, size_t *capacity_samples);
int dabgpu_receiver_reset(dabgpu_receiver *rx);
/* PRS slot = nb_fft samples from sample prs_sample of the current staging buffer (m_correlation_time_buffer[nb_null_period ...]); asynchronous */
int dabgpu_receiver_submit_sync(dabgpu_receiver *rx, const dabgpu_sync_cfg *cfg, size_t prs_sample);
/* blocks until the record of the last submit_sync is in host memory: freq_coarse / freq_fine / is_found_coarse after the synchroniser,
 * fine_time_offset, sync_valid (0: the caller resets, :529-532); h_impulse / h_freq_response [nb_fft] dB, may be NULL */
int dabgpu_receiver_wait_sync(dabgpu_receiver *rx, dabgpu_sync_state *out, float *h_impulse, float *h_freq_response);
/* the frame = samples-per-frame samples from sample frame_sample of the current staging buffer (nb_null_period + fine_time_offset); asynchronous:
 * upload, demodulation, fine-frequency update with `beta`, decode, results to host memory.  *generation counts the frames submitted. */
int dabgpu_receiver_submit_frame(dabgpu_receiver *rx, size_t frame_sample, float fine_freq_update_beta, int want_views, int tie_rule,
                                 uint64_t *generation);
/* The same in two calls, for two threads (what the OFDM_Demod mirror does): submit_demod enqueues the upload, the demodulation, the fine-frequency
 * update and the copies of what the host reads, and returns; submit_decode(generation) -- once per frame, in the order of the generations --
 * waits on the HOST until that frame is demodulated and then enqueues its decode.  No stream waits for another on the device (several receivers of
 * a process share hardware queues: a queue whose head waits holds back the other receiver's work behind it), and the thread that frames the stream
 * issues half the runtime calls.  At most 4 frames may be between the two calls (the decode of frame g reads the history slot frame g + 4 overwrites:
 * DABGPU_ERR_NOT_READY); dabgpu_receiver_wait_frame(g) only after submit_decode(g) has returned.  Replaces the same hand-over as above. */
int dabgpu_receiver_submit_demod(dabgpu_receiver *rx, size_t frame_sample, float fine_freq_update_beta, int want_views, uint64_t *generation);
int dabgpu_receiver_submit_decode(dabgpu_receiver *rx, uint64_t generation, int tie_rule);
int dabgpu_receiver_wait_frame(dabgpu_receiver *rx, uint64_t generation, dabgpu_receiver_frame *out);

/* ==================================================================================================
 * Transmission modes II, III and IV (geometries of src/ofdm/dab_ofdm_params_ref.cpp:11-60).
 * The same demodulation pipeline, frame-aligned batches, through a size-generic kernel (FFT 512 / 256 / 1024 with the mode I
 * butterflies and twiddle rule; PLL with apply_pll's scalar tail for symbol periods that are not a multiple of 4).  The DAB
 * layer above the soft bits exists for mode I only in the reference (fic_decoder.cpp:61-72), and here.
 */
/* out9 = {nb_frame_symbols, nb_symbol_period, nb_null_period, nb_fft, nb_cyclic_prefix, nb_data_carriers,
 *         samples per frame (symbols + NULL), soft bits per symbol, soft bits per frame}; host only */
int dabgpu_get_ofdm_params(int transmission_mode, int *out9);
/*   d_iq   [n_frames][samples per frame] complex float, layout as for mode I (symbols, PRS first, then the NULL symbol)
 *   d_bits [n_frames][soft bits per frame] int8, 16-byte aligned;  d_cp_corr [n_frames][nb_frame_symbols] complex float, may be NULL
 *   d_fft  [n_frames][nb_frame_symbols + 1][nb_fft] complex float, may be NULL.  transmission_mode 1 is accepted too
 *   (cross-check of the two kernels). */
int dabgpu_ofdm_demod_frames_mode(dabgpu_ctx *ctx, int transmission_mode, const float *d_iq, size_t n_frames,
                                  const float *d_freq_offset, int8_t *d_bits, float *d_cp_corr, float *d_fft, int symbols_per_block,
                                  void *stream);
int dabgpu_ofdm_phase_update_mode(dabgpu_ctx *ctx, int transmission_mode, const float *d_cp_corr, size_t n_frames,
                                  float fine_freq_update_beta, float *d_total_phase, float *d_fine_freq, void *stream);
/* PRS synchronisation (dabgpu_ofdm_sync) for any mode: d_prs_syms = nb_fft samples per stream; d_impulse / d_freq_response
 * [n_streams][nb_fft].  The PRS spectrum of the mode (get_DAB_PRS_reference, src/ofdm/dab_prs_ref.cpp:140-195) and the
 * coarse-sync reference are built on the context at first use. */
int dabgpu_ofdm_sync_mode(dabgpu_ctx *ctx, int transmission_mode, const float *d_prs_syms, size_t n_streams, size_t stride_samples,
                          const dabgpu_sync_cfg *cfg, dabgpu_sync_state *d_states, float *d_impulse, float *d_freq_response, void *stream);
/* single-stream host-buffer forms used by the OFDM_Demod mirror class in modes II-IV (h_fft [nb_frame_symbols+1][nb_fft]) */
int dabgpu_ofdm_demod_stream_frame_sync_mode(dabgpu_ctx *ctx, int transmission_mode, const float *h_iq, float freq_coarse,
                                             float *h_freq_fine, float fine_freq_update_beta, int8_t *h_bits, float *h_total_phase,
                                             float *h_fft);
int dabgpu_ofdm_sync_host_sync_mode(dabgpu_ctx *ctx, int transmission_mode, const float *h_prs_sym, const dabgpu_sync_cfg *cfg,
                                    dabgpu_sync_state *h_state, float *h_impulse, float *h_freq_response);

/* ==================================================================================================
 * Unsynchronised front end on the device: a bank of n independent receivers whose state between
 * calls -- signal level, NULL-search flags, circular NULL buffer, correlation window, frame buffer, frequency
 * offsets, counters (src/ofdm/ofdm_demodulator.h:131-176) -- lives in HBM.  One dabgpu_stream_bank_process() is one
 * OFDM_Demod::Process(block) for every stream (src/ofdm/ofdm_demodulator.cpp:235-275): L1 signal average (:934-950),
 * NULL power-dip search (:291-347), NULL+PRS read (:349-358), coarse/fine sync (:360-548), symbol read (:550-577),
 * frame demodulation and the fine-frequency loop (:581-639), Reset on a failed impulse-peak test (:277-289,:529-532).
 * Results equal the OFDM_Demod mirror class fed the same blocks, stream by stream, byte for byte.
 */
typedef struct {                      /* OFDM_Demod_Config, src/ofdm/ofdm_demodulator.h:24-45 */
    float signal_l1_update_beta;      /* 0.95 */
    int   signal_l1_nb_samples;       /* 100  */
    int   signal_l1_nb_decimate;      /* 5    */
    float thresh_null_start;          /* 0.35 */
    float thresh_null_end;            /* 0.75 */
    dabgpu_sync_cfg sync;
} dabgpu_stream_cfg;
void dabgpu_stream_cfg_default(dabgpu_stream_cfg *cfg);

typedef struct {
    int32_t state;                    /* OFDM_Demod::State, ofdm_demodulator.h:50-56 */
    float   signal_l1_average;
    float   freq_coarse, freq_fine;
    int32_t is_found_coarse;
    int32_t fine_time_offset;
    int32_t total_frames_read, total_frames_desync;
} dabgpu_stream_status;

typedef struct dabgpu_stream_bank dabgpu_stream_bank;
/* cfg NULL = the reference's defaults. Device memory: 1.64 MB per stream. */
int dabgpu_stream_bank_create(dabgpu_ctx *ctx, size_t n_streams, const dabgpu_stream_cfg *cfg, dabgpu_stream_bank **out);
/* the same for transmission mode 1..4 (get_DAB_OFDM_params, src/ofdm/dab_ofdm_params_ref.cpp:11-60): frames are
 * dabgpu_get_ofdm_params(mode)[8] soft bits each, ring / block limits scale with the mode's frame length; modes II-IV run the
 * size-generic demodulation kernel (dabgpu_ofdm_demod_frames_mode) */
int dabgpu_stream_bank_create_mode(dabgpu_ctx *ctx, int mode, size_t n_streams, const dabgpu_stream_cfg *cfg, dabgpu_stream_bank **out);
void dabgpu_stream_bank_destroy(dabgpu_stream_bank *bank);
int dabgpu_stream_bank_reset(dabgpu_stream_bank *bank, void *stream);     /* every stream back to its constructed state */
/*
 *   d_iq        stream s's block = n_samples complex float at d_iq + 2*s*stream_stride_samples (device, 8-byte aligned)
 *   d_bits      [n_streams][max_frames_per_stream][230400] int8: frame j completed by stream s during this call
 *   d_n_frames  [n_streams] out: frames completed during this call (may be NULL)
 *   max_frames_per_stream >= n_samples / 191400 + 2
 * Launches on `stream`; synchronises with it (the number of rounds depends on the data).
 */
int dabgpu_stream_bank_process(dabgpu_stream_bank *bank, const float *d_iq, size_t stream_stride_samples, size_t n_samples,
                               int8_t *d_bits, size_t max_frames_per_stream, int32_t *d_n_frames, void *stream);
/* The same from blocks still in a capture format (dabgpu_iq_format): d_raw holds n_samples IQ pairs per stream, stream s at
 * byte offset s * stream_stride_samples * sample_bytes.  raw_u8 / raw_s8 / raw_s16l (and wav PCM8 / PCM16 / float32) blocks are
 * read by the bank's kernels themselves -- level windows, NULL search, buffering and the demodulator's frame tails dequantise
 * on the fly with the reader arithmetic, 2-4 bytes per sample instead of 8 and no conversion pass; any other format is first
 * dequantised (dabgpu_iq_convert) into bank-owned scratch (then d_raw and the byte offset between streams must be multiples of 16). */
int dabgpu_stream_bank_process_raw(dabgpu_stream_bank *bank, const void *d_raw, int format, size_t stream_stride_samples,
                                   size_t n_samples, int8_t *d_bits, size_t max_frames_per_stream, int32_t *d_n_frames, void *stream);
/*
 * Retained blocks: dabgpu_stream_bank_process_raw for callers that keep a block's device memory valid and unchanged until the NEXT call
 * has returned -- a ring of device buffers (dabgpu_ingest_* with depth >= 2 is one).  OFDM_Demod::Process copies every sample it is
 * handed into its frame buffer (ofdm_demodulator.cpp:550-577); the bank already reads completed frames where they lie and copies only
 * the unfinished frame at a block's end; here that copy goes too: the next call's demodulator reads the frame's head from d_prev_raw,
 * its tail from d_raw.  Same frames, same state, bit for bit.
 *   d_prev_raw  the block of the previous retained call (same format and stride), NULL at the first call / after a release or reset
 *   format      one the kernels read directly: raw_f32l, raw_u8, raw_s8, raw_s16l, wav PCM8 / PCM16 / float32; mode I banks (banks of
 *               other modes process the call like dabgpu_stream_bank_process_raw)
 * A block shorter than a frame first copies what the previous block still holds (a frame then spans more than two blocks).
 * After a retained call the bank refuses any other process call until dabgpu_stream_bank_release (copies the carried samples out of the
 * last block, which may then be freed) or dabgpu_stream_bank_reset.
 */
int dabgpu_stream_bank_process_retained(dabgpu_stream_bank *bank, const void *d_raw, int format, size_t stream_stride_samples,
                                        size_t n_samples, const void *d_prev_raw, int8_t *d_bits, size_t max_frames_per_stream,
                                        int32_t *d_n_frames, void *stream);
int dabgpu_stream_bank_release(dabgpu_stream_bank *bank, const void *d_prev_raw, int format, size_t stream_stride_samples, void *stream);

/* Ring form for a device-resident pipeline: a completed frame of stream s is written to slot (frames demodulated so far) mod
 * hist_frames of d_hist [n_streams][hist_frames][230400] -- the per-ensemble frame-history ring dabgpu_fic_decode_ring /
 * dabgpu_msc_decode_ring read -- and d_newest_slot[s] receives that slot, or -1 when the stream completed no frame in this
 * call.  At most 191400 samples per call (so that a stream completes at most one frame); format must be one the kernels read
 * directly (raw_f32l, raw_u8, raw_s8, raw_s16l, wav PCM8/PCM16/float32). */
int dabgpu_stream_bank_process_ring(dabgpu_stream_bank *bank, const void *d_raw, int format, size_t stream_stride_samples,
                                    size_t n_samples, int8_t *d_hist, int hist_frames, int32_t *d_newest_slot, void *stream);
/* the same with the MSC part of every frame stored in `bits_layout` (DABGPU_BITS_NATURAL / DABGPU_BITS_MSC_CLASSED, see
 * dabgpu_ofdm_demod_frames_history; mode I banks) -- read the rings with dabgpu_fic_decode_ring + dabgpu_msc_decode_ring_layout */
int dabgpu_stream_bank_process_ring_layout(dabgpu_stream_bank *bank, const void *d_raw, int format, size_t stream_stride_samples,
                                           size_t n_samples, int8_t *d_hist, int hist_frames, int32_t *d_newest_slot,
                                           int bits_layout, void *stream);
/* The ring form with retained blocks (dabgpu_stream_bank_process_retained's contract: d_raw stays valid and unchanged until the NEXT call has
 * returned; d_prev_raw = the previous call's block, NULL at the first call / after a release or reset): the unfinished frame at the end of a
 * block stays where it is and the next call's demodulator reads it there -- in the ring form, whose blocks are shorter than a frame, the copy
 * it saves is half a frame per stream and call on average (and a u8 -> complex float expansion).  A frame that began in the last samples of a
 * block and cannot complete in the next one has those few samples copied at the start of the next call.  Same frames, same state, bit for bit. */
int dabgpu_stream_bank_process_ring_retained(dabgpu_stream_bank *bank, const void *d_raw, int format, size_t stream_stride_samples,
                                             size_t n_samples, const void *d_prev_raw, int8_t *d_hist, int hist_frames,
                                             int32_t *d_newest_slot, int bits_layout, void *stream);
/* snapshot of every stream's getters (GetState, GetSignalAverage, Get*FrequencyOffset, ...) into host memory; synchronous */
int dabgpu_stream_bank_status(dabgpu_stream_bank *bank, dabgpu_stream_status *h_status, void *stream);

/* ==================================================================================================
 * DAB+ outer code on the device: AAC_Frame_Processor between the channel decoder's bytes and the AAC
 * access units (src/dab/audio/aac_frame_processor.cpp:127-361): super-frame acquisition on the fire code, collection of 5
 * logical frames, RS(120,110) decoding of the interleaved columns (src/dab/algorithms/reed_solomon_decoder.cpp), fire code
 * of the corrected super frame, header walk and access-unit CRCs, re-acquisition after 10 failed super frames.  One
 * wavefront per stream (an (ensemble, sub-channel) pair); acquisition state and the super frame under collection stay in
 * HBM between calls.  Logical frames of 11..1536 bytes (every DAB+ sub-channel up to 512 kbit/s).
 */
typedef struct {
    int32_t rs_failed_index;        /* -1, or the first uncorrectable RS codeword (OnRSError, :336-341): super frame dropped */
    int32_t rs_corrected;           /* symbols reported corrected by the codewords decoded */
    int32_t firecode_ok;            /* fire code of the corrected super frame (:206-209) */
    int32_t header_valid;           /* OnSuperFrameHeader fired; the fields below are meaningful */
    int32_t descriptor;             /* byte 2: rfa | dac_rate | sbr_flag | aac_channel_mode | ps_flag | mpeg_surround_config(3) */
    int32_t num_aus;
    int32_t au_start[8];            /* byte offsets au_start[0..num_aus] inside the super frame (:266-283) */
    int32_t au_walk_stopped_at;     /* -1, or the access unit whose bounds test ended the walk (:291-297) */
    uint32_t au_crc_ok_mask;        /* bit i: OnAccessUnit(i) fired (CRC passed); other walked units: OnAccessUnitCRCError */
    int32_t frame_index;            /* index inside this call of the logical frame that completed the super frame */
    uint32_t firecode_rx_calc;      /* received << 16 | calculated fire code of the corrected super frame (OnFirecodeError arguments) */
    uint16_t au_crc_calc[6];        /* calculated CRC of every walked access unit (OnAccessUnitCRCError arguments) */
    uint16_t reserved[2];
} dabgpu_superframe_result;

typedef struct dabgpu_dabplus_bank dabgpu_dabplus_bank;
int dabgpu_dabplus_bank_create(dabgpu_ctx *ctx, size_t n_streams, dabgpu_dabplus_bank **out);
void dabgpu_dabplus_bank_destroy(dabgpu_dabplus_bank *bank);
int dabgpu_dabplus_bank_reset(dabgpu_dabplus_bank *bank, void *stream);
/*
 * n_frames x AAC_Frame_Processor::Process per stream.  Logical frame f of stream s = d_frame_bytes[s] bytes at
 * d_frames + d_stream_offsets[s] + f*frame_stride_bytes (the layout dabgpu_msc_decode_frames writes: offset of the
 * sub-channel inside one CIF record, stride of a CIF record).
 *   d_superframes [n_streams][max_superframes][superframe_stride_bytes]: every super frame ATTEMPTED in this call (5 frames
 *                 collected), corrected where RS succeeded;  d_results the matching records
 *   d_counts      [n_streams][4]: super frames attempted, logical frames dropped while waiting for a valid fire code,
 *                 received << 16 | calculated fire code of the last dropped frame, logical frames collected so far
 *   max_superframes >= ceil(n_frames / 5), superframe_stride_bytes >= 5 * frame bytes
 * Limits, reported per stream instead of being skipped in silence: a stream whose logical frames exceed 1536 bytes (the reference
 * takes any N >= 11, aac_frame_processor.cpp:129-137) is not processed and gets d_counts[s][1] = -1; a stream whose
 * 5 * frame bytes exceed superframe_stride_bytes gets d_counts[s][1] = -2.  (dabgpu_dabplus_process_frame_host_sync returns
 * DABGPU_ERR_UNSUPPORTED for the first case.)
 */
int dabgpu_dabplus_bank_process(dabgpu_dabplus_bank *bank, const uint8_t *d_frames, const uint64_t *d_stream_offsets,
                                size_t frame_stride_bytes, const uint32_t *d_frame_bytes, int n_frames, uint8_t *d_superframes,
                                size_t superframe_stride_bytes, dabgpu_superframe_result *d_results, int max_superframes,
                                int32_t *d_counts, void *stream);
/* the same, skipping every stream whose flag d_active[s / streams_per_flag] is negative (e.g. the newest-slot array of
 * dabgpu_stream_bank_process_ring with streams ordered ensemble-major: streams_per_flag = DAB+ sub-channels per ensemble) */
int dabgpu_dabplus_bank_process_masked(dabgpu_dabplus_bank *bank, const uint8_t *d_frames, const uint64_t *d_stream_offsets,
                                       size_t frame_stride_bytes, const uint32_t *d_frame_bytes, int n_frames, uint8_t *d_superframes,
                                       size_t superframe_stride_bytes, dabgpu_superframe_result *d_results, int max_superframes,
                                       int32_t *d_counts, const int32_t *d_active, int streams_per_flag, void *stream);
/* one Process(buf) of a one-stream bank with host buffers (the AAC_Frame_Processor mirror class); h_superframe [5*n_bytes] */
int dabgpu_dabplus_process_frame_host_sync(dabgpu_dabplus_bank *bank, const uint8_t *h_frame, uint32_t n_bytes, int *superframe_done,
                                           int *firecode_wait_failed, uint32_t *firecode_wait_rx_calc /* may be NULL */,
                                           dabgpu_superframe_result *h_result, uint8_t *h_superframe);

/* ==================================================================================================
 * Data formats either side of the path.
 *
 * IQ input: the reference's readers (examples/app_helpers/app_iq_readers.h:17-159, app_wav_reader.h:257-470)
 * turn a byte stream into std::complex<float>; here the raw bytes go to the device as they are and one
 * kernel applies the same arithmetic:
 *   quantised integers (app_iq_readers.h:19-44,79-84): (float(v) - BIAS) * (1.0f/MAX_AMPLITUDE), BIAS = 0 and
 *     MAX_AMPLITUDE = float(max) for signed T; BIAS = MAX_AMPLITUDE = float(max/2) + 0.5f for unsigned T;
 *   raw_f32*: bit copy (after the byte swap); raw_f64*: (float)double;
 *   wav: PCM8 (v-127.5f)*(1/127.5f), PCM16 v*(1/32767.f), PCM24 v*(1/8388607.f), PCM32 v*(1/float(INT32_MAX)),
 *     IEEE float 32/64, G.711 A-law / mu-law (app_wav_reader.h:271-456).
 * Soft/hard bit files: examples/app_helpers/app_viterbi_convert_block.h:12-44 (bit i of a byte, LSB first,
 * <-> soft bit +127 / -127; hard = soft >= 0).
 */
typedef enum {
    DABGPU_IQ_RAW_U8 = 0, DABGPU_IQ_RAW_S8,
    DABGPU_IQ_RAW_S16L, DABGPU_IQ_RAW_S16B, DABGPU_IQ_RAW_U16L, DABGPU_IQ_RAW_U16B,
    DABGPU_IQ_RAW_S32L, DABGPU_IQ_RAW_S32B, DABGPU_IQ_RAW_U32L, DABGPU_IQ_RAW_U32B,
    DABGPU_IQ_RAW_F32L, DABGPU_IQ_RAW_F32B, DABGPU_IQ_RAW_F64L, DABGPU_IQ_RAW_F64B,
    DABGPU_IQ_WAV_PCM8, DABGPU_IQ_WAV_PCM16, DABGPU_IQ_WAV_PCM24, DABGPU_IQ_WAV_PCM32,
    DABGPU_IQ_WAV_F32, DABGPU_IQ_WAV_F64, DABGPU_IQ_WAV_ALAW, DABGPU_IQ_WAV_MULAW,
    DABGPU_IQ_NB_FORMATS
} dabgpu_iq_format;

/* mode strings of app_iq_readers.h:107-113 ("raw_u8" ... "raw_f64b") -> dabgpu_iq_format; "wav" and unknown
 * strings return -1 (a wav file's format comes from dabgpu_wav_parse_header). Host only, needs no device. */
int dabgpu_iq_format_from_mode(const char *mode);
/* bytes of one IQ sample (two components) in the given format, 0 for an invalid format. Host only. */
size_t dabgpu_iq_format_sample_bytes(int format);

typedef struct {
    int32_t  iq_format;             /* dabgpu_iq_format of the sample data (DABGPU_IQ_WAV_*) */
    uint16_t audio_format;          /* 1 PCM, 3 IEEE float, 6 A-law, 7 mu-law (after resolving WAVE_FORMAT_EXTENSIBLE) */
    uint16_t total_channels;
    uint32_t samples_per_second;
    uint32_t average_bytes_per_second;
    uint16_t data_block_align_bytes;
    uint16_t bits_per_sample;
    uint32_t data_chunk_size;       /* bytes */
    uint64_t data_chunk_offset;     /* byte offset of the sample data from the start of the file */
} dabgpu_wav_header;
/* wav_read_header (app_wav_reader.h:107-255) over the first n_bytes of a file held in memory; accepts what it
 * accepts (fmt chunk of 16/18/40 bytes, 1 or 2 channels, fact chunk required for non-PCM, non-data chunks skipped)
 * and rejects what it or WavFileReader's constructor rejects (DABGPU_ERR_INVALID_ARG + dabgpu_last_error()).
 * get_iq_file_reader_from_mode_string additionally requires total_channels == 2 (app_iq_readers.h:122-126): callers
 * feeding dabgpu_iq_convert check that. Host only, needs no device. */
int dabgpu_wav_parse_header(const uint8_t *bytes, size_t n_bytes, dabgpu_wav_header *out);

/* d_raw: n_samples IQ pairs in `format` on the device (16-byte aligned); d_iq: n_samples interleaved (re, im) floats */
int dabgpu_iq_convert(dabgpu_ctx *ctx, const void *d_raw, int format, size_t n_samples, float *d_iq, void *stream);
int dabgpu_iq_convert_host_sync(dabgpu_ctx *ctx, const void *h_raw, int format, size_t n_samples, float *h_iq);

/* convert_viterbi_bits_to_bytes (app_viterbi_convert_block.h:28-44): n_bytes output bytes from 8*n_bytes soft bits */
int dabgpu_soft_bits_to_hard_bytes(dabgpu_ctx *ctx, const int8_t *d_bits, size_t n_bytes, uint8_t *d_bytes, void *stream);
/* convert_viterbi_bytes_to_bits (app_viterbi_convert_block.h:12-26) */
int dabgpu_hard_bytes_to_soft_bits(dabgpu_ctx *ctx, const uint8_t *d_bytes, size_t n_bytes, int8_t *d_bits, void *stream);
int dabgpu_soft_bits_to_hard_bytes_host_sync(dabgpu_ctx *ctx, const int8_t *h_bits, size_t n_bytes, uint8_t *h_bytes);
int dabgpu_hard_bytes_to_soft_bits_host_sync(dabgpu_ctx *ctx, const uint8_t *h_bytes, size_t n_bytes, int8_t *h_bits);

/* ------------------------------------------------------------------------------------------------------------------------------
 * Ingest pipe: the host -> device hand-over of capture bytes.  Replaces the reader thread -> OFDM_Demod::Process hand-over
 * of examples/app_helpers/app_ofdm_blocks.h:45-58 and the memcpy of OFDM_Demod::ReadSymbols (src/ofdm/ofdm_demodulator.cpp:550-577).
 * A ring of `depth` PINNED host buffers with device twins and a copy stream of its own:
 *     dabgpu_ingest_acquire(pipe, &h)            fill h (e.g. fread straight into it; capture format, 2 B per sample for raw_u8)
 *     dabgpu_ingest_submit(pipe, bytes, &d)      asynchronous copy on the pipe's copy stream; d = device twin
 *     dabgpu_ingest_wait(pipe, d, s)             stream s waits for the copy into d (right before the kernels that read it: a batch may be copied ahead)
 *     ... enqueue kernels reading d on s (dabgpu_ofdm_demod_frames_raw, dabgpu_stream_bank_process_raw, ...)
 *     dabgpu_ingest_consumed(pipe, d, s)         d may be overwritten once those kernels have run
 * With depth >= 2 the copy of batch k + 1 overlaps the demodulation of batch k; acquire blocks only while the ring is full.
 * One pipe = one producer thread. */
typedef struct dabgpu_ingest dabgpu_ingest;
int dabgpu_ingest_create(dabgpu_ctx *ctx, size_t buffer_bytes, int depth, dabgpu_ingest **out);
void dabgpu_ingest_destroy(dabgpu_ingest *pipe);
int dabgpu_ingest_acquire(dabgpu_ingest *pipe, void **h_buffer);
int dabgpu_ingest_submit(dabgpu_ingest *pipe, size_t bytes, void **d_buffer);
int dabgpu_ingest_wait(dabgpu_ingest *pipe, const void *d_buffer, void *compute_stream);
int dabgpu_ingest_consumed(dabgpu_ingest *pipe, const void *d_buffer, void *compute_stream);

#ifdef __cplusplus
}
#endif
#endif /* DABGPU_H */
