// dabgpu_internal.h -- shared between the translation units of libdabgpu.so (not installed)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <mutex>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_host_logic.h"

struct dabgpu_ctx {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    std::vector<float> prs;          // host copy, 2*2048
    std::vector<int> mapper;         // host copy, 1536
    float* d_tw = nullptr;           // 2048 x (cos, -sin)
    uint16_t* d_inv_map = nullptr;   // carrier -> de-interleaved position
    float* d_prs = nullptr;          // PRS spectrum
    float* d_prs_time_ref = nullptr; // conj(IFFT(relative_phase(PRS))), coarse-sync reference
    struct dabgpu_vit_tables* d_vit_tables = nullptr;
    int vit_mapping = 0;             // DABGPU_VIT_MAP_* (dabgpu_viterbi_set_mapping)
    // symbols_per_block = 0 of the mode I demodulator: what dabgpu_ofdm_tune measured on this device, per size bucket (ceil log2 of the
    // batch) and kernel variant (loader, soft-bit layout, phase tail); the data path only looks it up (dabgpu_abi.hip)
    struct spb_choice { int bucket; int variant; int spb; };
    std::vector<spb_choice> spb_cache;
    int* d_mode_mapper[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // carrier mappers of modes II-IV, built on first use
    int* d_mode_inv_map[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // their inverses (ofdm_wave512.hip)
    float* d_mode_prs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};    // PRS spectra and coarse-sync references of modes II-IV
    float* d_mode_prs_time_ref[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<void*> scratch;      // grow-only device scratch slots
    std::vector<size_t> scratch_bytes;
    std::vector<void*> parked;       // outgrown slots a captured graph may still address (dabgpu_scratch); freed with the context
    bool captured_once = false;      // a capturable entry point of this context has run under hipStreamBeginCapture
    // Host-side entry points (*_host_sync, msc_stream_*, dabplus_process_frame_host_sync) share the context's stream and scratch
    // slots: they serialise on this lock, so decoder objects living on different threads (BasicThreadPool workers,
    // src/basic_radio/basic_radio.cpp:51-60) may share one context.  The batch entry points (device pointers + caller's stream)
    // take no lock: one thread per context, or the caller's own exclusion.
    std::recursive_mutex host_mu;
    // pinned staging ring for small host -> device copies whose source does not outlive the call (descriptor tables, a caller's
    // CIF): copy into a pinned slot, asynchronous DMA from there, an event marks the slot reusable
    struct stage_slot { void* h = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool pending = false; std::mutex mu; };
    stage_slot stage[8];
    unsigned stage_next = 0;
    std::mutex stage_mu;
    // small tables that are a function of the call's arguments only (the sub-channel plans of a multiplex): what was last uploaded where
    // and on which stream -- a call that would upload the same bytes to the same place on the same stream uploads nothing
    // (dabgpu_stage_h2d_cached).  Steady-state decode calls then contain kernel launches only, which is what lets a caller capture them
    // in a HIP graph.
    struct table_copy { void* d = nullptr; hipStream_t s = nullptr; std::vector<unsigned char> bytes; };
    table_copy tables[2];
    std::mutex tables_mu;
};
#define DABGPU_HOST_LOCK(ctx) std::lock_guard<std::recursive_mutex> dabgpu_host_lock_(ctx->host_mu)
// asynchronous host -> device copy on `s` that has consumed h_src when it returns (h_src may be freed or overwritten at once)
extern "C" int dabgpu_stage_h2d(dabgpu_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
// the same for table `which` (0: sub-channel plans, 1: lane table) of the context, skipped when nothing changed (see dabgpu_ctx::tables)
extern "C" int dabgpu_stage_h2d_cached(dabgpu_ctx* c, int which, void* d_dst, const void* h_src, size_t bytes, hipStream_t s);

int dabgpu_check_hip(hipError_t e, const char* what);
// Every entry point that launches, allocates or copies first makes the context's device current on the calling thread (a worker
// thread of a one-process multi-GPU host starts on device 0) -- checked: a failed hipSetDevice is the call's status.
int dabgpu_bind_device(const dabgpu_ctx* c);
#define DABGPU_BIND(ctx) do { const int dabgpu_bind_st_ = dabgpu_bind_device(ctx); if (dabgpu_bind_st_) return dabgpu_bind_st_; } while (0)
// flags of the events host threads wait on (receiver pipeline, frame session, receiver bank): DABGPU_EVENT_WAIT=block makes hipEventSynchronize SLEEP
// instead of spinning -- threads that wait for the device then cost no CPU (a container's CPU quota is shared by every waiting thread of a many-receiver
// process), at ~20-50 us more wake-up latency; =spin is the runtime's default.  Unset: `bank_default` for the receiver bank's events, spin elsewhere.
unsigned dabgpu_wait_event_flags(bool bank_default_block);
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out);
// the same from a capturable entry point launching on `user` (HIP graphs: see the definition)
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out, hipStream_t user);
// device PRS spectrum / coarse-sync time reference of a transmission mode (mode I: the context's own tables)
int dabgpu_mode_sync_tables(dabgpu_ctx* c, int mode, const float** d_prs, const float** d_prs_time_ref);

// per-stream work item of a stream bank round (ofdm_stream.hip -> ofdm_demod.hip)
struct dabgpu_frame_desc {
    int slot;               // output slot of the completed frame, < 0: nothing to demodulate
    int split;              // samples [0, split) come from the stream's frame buffer (even)
    long long tail_off;     // sample offset inside the stream's current block of frame sample `split`
    // retained blocks (dabgpu_stream_bank_process_retained): frame samples [carry_dst, carry_end) (even bounds, inside [0, split)) are read
    // from the stream's PREVIOUS block, frame sample n at block sample carry_off + n; carry_end = 0: none
    int carry_dst, carry_end;
    long long carry_off;
};

extern "C" hipError_t dabgpu_launch_ofdm_demod(const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                               float* d_fft, float* d_dqpsk, const float* d_tw, const uint16_t* d_inv_map,
                                               int n_frames, int sym_per_chunk, size_t bits_frame_stride,
                                               const dabgpu_frame_desc* d_desc, const void* d_tail, size_t tail_stride,
                                               int classed, hipStream_t stream, float* d_total_phase = nullptr, float* d_fine_freq = nullptr,
                                               float beta = 0.0f, const void* d_prev_tail = nullptr /* stream banks, retained blocks */,
                                               size_t frame_stride_samples = 0 /* 0 = 196608, frame after frame */,
                                               dabgpu_sync_state* d_sync = nullptr /* frame k starts prs_offset + d_sync[k].fine_time_offset samples into its
                                                                                      slice, PLL offset and fine-frequency word are the record's */,
                                               int prs_offset = 0);
extern "C" hipError_t dabgpu_launch_ofdm_phase(const float* d_cp_corr, int n_frames, float beta, float* d_total_phase,
                                               float* d_fine_freq, int fine_freq_stride, const dabgpu_frame_desc* d_desc, int n_sym, int n_fft,
                                               hipStream_t stream);

// size-generic demodulation (ofdm_modes.hip); d_desc != nullptr = stream bank round (frame = stream, split input)
int dabgpu_launch_ofdm_demod_mode(dabgpu_ctx* c, int mode, const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                  float* d_fft, int n_frames, int symbols_per_block, const dabgpu_frame_desc* d_desc, const void* d_block,
                                  size_t block_stride, hipStream_t s);

// register-resident demodulation of modes II / IV (ofdm_wave512.hip), same arguments
int dabgpu_launch_ofdm_demod_wave(dabgpu_ctx* c, int mode, const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                  int n_frames, int symbols_per_block, const dabgpu_frame_desc* d_desc, const void* d_block,
                                  size_t block_stride, hipStream_t s);

// ---- channel decode ----
typedef dabgpu_codeword dabgpu_cw_desc;
typedef dabgpu_codeword_result dabgpu_cw_result;
#define DABGPU_CW_LANE_MAPPED 0x80000000u      // internal flag bit of dabgpu_codeword.flags (set by msc_build_descs_kernel)
// lane-per-codeword decoder (viterbi_lanes.hip): a GROUP = up to 64 codewords with one puncturing schedule.
// Symbol area of a group: the codewords' KEPT soft bits only, transposed -- row j, lane L = input bytes 4 j .. 4 j + 3 of lane L's
// codeword (after the time de-interleaver, -128 clamped to -127).  The trellis kernel de-punctures with wave-uniform selectors.
struct dabgpu_vit_group {
    uint32_t first, stride, count;  // lane L decodes descs[first + L * stride], L < count
    uint32_t n_steps;               // trellis steps incl. tail
    uint32_t alloc_steps;           // rows of the group's decision area (dabgpu_vit_alloc_steps)
    uint32_t seg_pi[4];
    uint32_t seg_steps[4];
    uint32_t in_rows;               // rows of the group's symbol area (dabgpu_vit_in_rows)
    uint64_t sched_off;             // entries into the schedule tables: (symbol row, v_perm_b32 selector) per trellis step
    uint64_t sym_off;               // dwords into the symbol scratch   [in_rows][64]
    uint64_t dec_off;               // dwords into the decision scratch [alloc_steps][64][2]
    int64_t res_delta;              // bytes added to &results[first + L * stride]: groups of ONE launch may report into different arrays
                                    // (the FIB groups of a frame decoded inside the MSC launch, dabgpu_decode_frames_layout)
};
extern "C" hipError_t dabgpu_launch_vit_groups_uniform(dabgpu_vit_group* d_groups, size_t n_cw, uint32_t n_steps,
                                                       const uint32_t* seg_pi, const uint32_t* seg_steps, hipStream_t stream);
// the same groups appended to another launch's: descriptor index, schedule entries, symbol / decision dwords and result bytes they start at
struct dabgpu_vit_group_base { uint32_t first; uint64_t sched_off, sym_off, dec_off; int64_t res_delta; };
extern "C" hipError_t dabgpu_launch_vit_groups_uniform_at(dabgpu_vit_group* d_groups, size_t n_cw, uint32_t n_steps, const uint32_t* seg_pi,
                                                          const uint32_t* seg_steps, dabgpu_vit_group_base base, hipStream_t stream);
// the two halves of dabgpu_launch_viterbi_lanes: the gather of `kind` (0 general, 1 ring of 4 CIFs, 2 the same in class order, 3 direct)
// over some groups, and the trellis over groups that were gathered
extern "C" hipError_t dabgpu_launch_vit_prep(int kind, const dabgpu_vit_group* d_groups, size_t n_groups, uint32_t max_in_rows,
                                             const dabgpu_cw_desc* d_descs, uint32_t* d_sym, uint32_t groups_per_sub, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_vit_trellis(const dabgpu_vit_group* d_groups, size_t n_groups, const dabgpu_cw_desc* d_descs, uint32_t* d_sym,
                                                uint32_t* d_dec, dabgpu_cw_result* d_results, int tie_rule, const struct dabgpu_vit_tables* d_tables,
                                                const uint2* d_sched, int octet, int n_cu, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_vit_groups_msc(dabgpu_vit_group* d_groups, const struct dabgpu_msc_plan* d_plans,
                                                   const uint64_t* d_lane_subs, int n_lane_sub, int n_sub, size_t n_ens,
                                                   uint32_t groups_per_sub, uint32_t sched_stride, hipStream_t stream);
// schedule tables (one per puncturing schedule of the call): sched_stride entries each.  The trellis kernels read entry t for the steps
// t < n_steps and prefetch up to DABGPU_VIT_SCHED_PREFETCH entries beyond the last step, so sched_stride >= n_steps + DABGPU_VIT_SCHED_PREFETCH;
// callers size the tables with dabgpu_vit_alloc_steps (>= n_steps + 6, static check below)
extern "C" hipError_t dabgpu_launch_vit_sched_uniform(uint2* d_sched, uint32_t sched_stride, const uint32_t* seg_pi, const uint32_t* seg_steps,
                                                      const struct dabgpu_vit_tables* d_tables, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_vit_sched_msc(uint2* d_sched, uint32_t sched_stride, const struct dabgpu_msc_plan* d_plans,
                                                  const uint64_t* d_lane_subs, int n_lane_sub, const struct dabgpu_vit_tables* d_tables,
                                                  hipStream_t stream);
extern "C" hipError_t dabgpu_launch_viterbi_lanes(const dabgpu_vit_group* d_groups, size_t n_groups, uint32_t max_in_rows,
                                                  const dabgpu_cw_desc* d_descs, uint32_t* d_sym, uint32_t* d_dec,
                                                  dabgpu_cw_result* d_results, int tie_rule, int ring4,
                                                  const struct dabgpu_vit_tables* d_tables, const uint2* d_sched, int octet, int n_cu,
                                                  uint32_t groups_per_sub /* MSC: groups[li * groups_per_sub + gq]; 0 = any order */,
                                                  hipStream_t stream);
// eight lanes per codeword over prepared groups and their symbol array (viterbi_octet.hip): one 512-thread workgroup per group
extern "C" hipError_t dabgpu_launch_viterbi_octet(const dabgpu_vit_group* d_groups, size_t n_groups, const dabgpu_cw_desc* d_descs,
                                                  const uint32_t* d_sym, uint32_t* d_dec, dabgpu_cw_result* d_results, int tie_rule,
                                                  const struct dabgpu_vit_tables* d_tables, const uint2* d_sched, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_viterbi(const dabgpu_cw_desc* d_descs, int n_cw, uint64_t* d_scratch,
                                            size_t scratch_words_per_wave, int n_waves, int max_out_bytes,
                                            dabgpu_cw_result* d_results, int tie_rule, const dabgpu_vit_tables* d_tables,
                                            hipStream_t stream, int n_first = 0 /* > 0: code words n_first .. n_cw - 1 report into d_results_rest[0 ..] */,
                                            dabgpu_cw_result* d_results_rest = nullptr);
extern "C" hipError_t dabgpu_launch_fic_build(dabgpu_cw_desc* d_descs, const int8_t* d_bits, size_t n_frames,
                                              size_t frame_stride, uint8_t* d_out, const int32_t* d_slots, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_msc_fic_build(dabgpu_cw_desc* d_descs, const int8_t* d_hist, size_t n_ens, size_t ens_stride,
                                                  int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* d_plans, int n_sub,
                                                  uint8_t* d_out, size_t out_ens_stride, int cif_out_bytes, const int32_t* d_slots,
                                                  int classed, const int8_t* d_fic_bits, uint8_t* d_fib_out, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_msc_build(dabgpu_cw_desc* d_descs, const int8_t* d_hist, size_t n_ens, size_t ens_stride,
                                              int hist_frames, int newest_frame_slot, const dabgpu_msc_plan* d_plans, int n_sub,
                                              uint8_t* d_out, size_t out_ens_stride, int cif_out_bytes, const int32_t* d_slots,
                                              int classed, hipStream_t stream);

// ---- frame session (dabgpu_decode_abi.hip) ----
// One receiver's decode state behind the single-stream classes: an 8-frame history of soft bits on the device, the FIC + MSC decode of
// every frame pushed, result slots in pinned host memory (include/dabgpu.h, "Frame session").  Frames arrive either from host memory
// (dabgpu_frame_session_push_frame) or -- the receiver pipeline, receiver.hip -- are demodulated straight into the history slot by a
// producer stream: dabgpu_session_reserve hands the slot out, dabgpu_session_commit chains the decode behind the producer's event.
struct dabgpu_frame_session {
    static constexpr int H = 8, R = 8;
    dabgpu_ctx* ctx = nullptr;
    bool owns_ctx = true;                           // false: a result store of a receiver-bank member (receiver_bank.hip): no history, no stream of its own
    int8_t* d_hist = nullptr;                       // [H][230400]
    // one device block per session, one pinned block per result slot: [4][96] FIB bytes | [4] FIC results | [4][n_sub] MSC results |
    // [4][cif_out] sub-channel bytes -- a frame's results reach the host in ONE copy (they were four; each is an operation on the stream
    // between two trellis launches)
    uint8_t* d_block = nullptr; size_t block_bytes = 0;
    uint8_t* d_fib = nullptr; dabgpu_codeword_result* d_fres = nullptr;       // (into d_block)
    uint8_t* d_msc = nullptr; dabgpu_codeword_result* d_mres = nullptr;
    std::vector<dabgpu_subchannel> subs;
    std::vector<uint32_t> sub_off, sub_n;           // byte offset / size of a sub-channel inside one CIF's output record
    uint32_t cif_out = 0;
    uint64_t next_gen = 0;                          // next generation to be committed (decode enqueued), in order
    uint64_t next_reserve = 0;                      // next generation to be reserved (>= next_gen: a producer may run ahead of the commits)
    struct slot {
        uint64_t gen = ~0ull; bool fic = false, pending = false;
        std::vector<dabgpu_subchannel> subs; std::vector<uint32_t> sub_off, sub_n; uint32_t cif_out = 0;
        uint8_t* h_block = nullptr; size_t h_block_cap = 0;                      // pinned, laid out like the session's d_block
        uint8_t* h_fib = nullptr; dabgpu_codeword_result* h_fres = nullptr;      // (into h_block) [4][96], [4]
        uint8_t* h_msc = nullptr; dabgpu_codeword_result* h_mres = nullptr;      // (into h_block) [4][cif_out], [4][n_sub]
        hipEvent_t done = nullptr;
        hipEvent_t ev_ready = nullptr, ev_copied = nullptr;                    // receiver pipeline: frame demodulated / its host copies made (producer stream)
        // receiver pipeline only (pinned, allocated at first use): the frame's soft bits, a few scalars of the producer, display views
        int8_t* h_bits = nullptr; float* h_aux = nullptr; float* h_fft = nullptr; float* h_dq = nullptr;
        size_t h_fft_cap = 0, h_dq_cap = 0;
    } slots[R];
    std::mutex mu;
};
// the history slot the next frame goes to (*d_frame_bits) and its result slot; waits (host) for the result slot's previous frame, makes
// `producer` wait (device) for the decode that still reads the history slot.  No frame is pushed yet: dabgpu_session_commit does that.
int dabgpu_session_reserve(dabgpu_frame_session* s, hipStream_t producer, uint64_t* gen, int8_t** d_frame_bits, dabgpu_frame_session::slot** sl);
// the frame reserved last is in its history slot once `ready` (recorded on the producer stream) has fired: the session's stream waits for
// it, copies `bits_bytes` soft bits to the slot's h_bits (0 = no copy), decodes, waits for `producer_done` (if given: copies the producer
// still makes to the slot's pinned buffers on its own stream, beside the decode) and records the slot's done event
int dabgpu_session_commit(dabgpu_frame_session* s, uint64_t gen, hipEvent_t ready, size_t bits_bytes, int decode, int decode_fic, int tie_rule,
                          hipEvent_t producer_done = nullptr);
void dabgpu_session_unreserve(dabgpu_frame_session* s, uint64_t gen);
// a session that only STORES results (slots filled by the receiver bank); ctx is borrowed
int dabgpu_frame_session_create_store(dabgpu_frame_session** out, dabgpu_ctx* ctx);
// result slot of a generation, waited for (DABGPU_ERR_NOT_READY: gone or never pushed); call with s->mu held
int dabgpu_session_slot(dabgpu_frame_session* s, uint64_t gen, dabgpu_frame_session::slot** out);

// ---- sync ----
extern "C" hipError_t dabgpu_launch_sync_init(const float* d_prs, const float* d_tw, float* d_prs_time_ref, int n_fft, hipStream_t stream);
extern "C" hipError_t dabgpu_launch_sync(const float* d_prs_syms, size_t stride_samples, int n_streams, const dabgpu_sync_cfg* cfg,
                                         dabgpu_sync_state* d_states, float* d_impulse, float* d_freq, const float* d_tw,
                                         const float* d_prs, const float* d_prs_time_ref, const int* d_active, int mode,
                                         hipStream_t stream);
extern "C" hipError_t dabgpu_launch_cif_deinterleave(const int8_t* d_ring, int n_bits, int n_slots, int newest_slot,
                                                     int8_t* d_out, hipStream_t stream);
