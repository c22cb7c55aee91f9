// receiver.hip -- the receiver pipeline of the C ABI (include/dabgpu.h, "Receiver pipeline"): one receiver's per-frame device work
// enqueued without a host wait in between, SURVEY P2 behind the OFDM_Demod mirror class.
//
// The reference overlaps three things per frame (src/ofdm/ofdm_demodulator.cpp:550-577, :581-639): the reader thread buffers frame k + 1
// and runs its PRS synchronisation while the pipeline threads demodulate frame k and the coordinator thread calls the observers, which
// decode it (src/basic_radio/basic_radio.cpp:41-65).  Here the same overlap is two device streams and events:
//   stream A (the receiver's own,      H2D PRS -> ofdm_sync_kernel -> D2H record           per frame, as soon as the PRS slot is buffered
//             top stream priority)     H2D frame -> ofdm_demod_kernel (soft bits straight into the frame session's history slot)
//                                      -> ofdm_phase_kernel on the device-resident frequency state -> event `ready`
//                                      -> D2H soft bits, scalars, display views -> event `copied`
//   stream B (the frame session's)     waits for `ready` -> FIC + MSC decode of the frame -> ONE D2H of the result block -> waits for `copied` -> done event
// The frequency state (m_freq_coarse_offset, m_freq_fine_offset, m_is_found_coarse_freq_offset) lives on the device: the synchroniser of
// frame k + 1 reads the fine-frequency word frame k's phase kernel wrote, in stream order, and never waits for frame k's decode.
// The host learns two things per frame and both late: the synchroniser's record (before it must know where the frame ends) and the
// frame's results (dabgpu_receiver_wait_frame, on whatever thread delivers frames).
#include <hip/hip_runtime.h>
#include <string.h>
#include <atomic>

#include "dabgpu.h"
#include "dabgpu_internal.h"
#include "receiver_bank.h"

namespace {
constexpr int STAGES = 3;

__global__ void rx_net_freq_kernel(const dabgpu_sync_state* __restrict__ st, float* __restrict__ net) {
    // :672 the PLL of a frame runs with m_freq_coarse_offset + m_freq_fine_offset
    if (threadIdx.x == 0) net[0] = st->freq_coarse + st->freq_fine;
}
}  // namespace

// the device record's head: the state, then the 4 scalars; the responses follow at REC_HEAD
static constexpr size_t REC_SMALL = (sizeof(dabgpu_sync_state) + 15) & ~(size_t)15, REC_HEAD = REC_SMALL + 4 * sizeof(float);

struct dabgpu_receiver {
    dabgpu_ctx* ctx = nullptr;               // tables of the mode
    hipStream_t a = nullptr;                 // stream A, at the device's highest stream priority: its launches are tens of microseconds long and the
                                             // host's frame position hangs on them; they must not queue behind another receiver's 0.2 ms trellis launch
    dabgpu_frame_session* ses = nullptr;     // history ring, decode, result slots + stream B
    int mode = 1;
    int geom[9] = {0};
    size_t stage_cap = 0;                    // samples
    float* h_stage[STAGES] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_free[STAGES] = {nullptr, nullptr, nullptr};
    bool stage_pending[STAGES] = {false, false, false};
    int cur = 0;
    float* d_prs = nullptr;                  // nb_fft samples of the PRS slot
    float* d_iq = nullptr;                   // one frame, aligned
    // one device record: [frequency state + last synchroniser record | 4 scalars | impulse response | coarse frequency response] -- the host reads
    // the state and the scalars after a frame in ONE copy (the head), the head and the responses after a synchronisation
    void* d_rec = nullptr;
    dabgpu_sync_state* d_state = nullptr;    // (into d_rec)
    float* d_small = nullptr;                // (into d_rec) [0] net offset of the frame, [2] sum of the cyclic-prefix angles
    float* d_corr = nullptr;
    float* d_imp = nullptr;                  // (into d_rec) impulse response | coarse frequency response, nb_fft floats each
    float* d_fft = nullptr; float* d_dq = nullptr;
    // pinned: the synchroniser's record and responses
    unsigned char* h_rec = nullptr;          // laid out like d_rec
    hipEvent_t sync_done = nullptr; bool sync_pending = false; bool sync_coarse = false;
    std::atomic<int> decode_fic{0};          // (written by set_subchannels, read where the decode is submitted: possibly another thread)
    // a BANKED receiver (receiver_bank.hip): the staging buffers above are its own, everything else is the bank's; submit_* post jobs
    dabgpu_rx_member* member = nullptr;
    int device = 0;
};

#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)

extern "C" void dabgpu_receiver_destroy(dabgpu_receiver* rx) {
    if (!rx) return;
    if (rx->member) {
        dabgpu_rx_bank_leave(rx->member);                            // (waits for the member's jobs: nothing reads the staging buffers afterwards)
        (void)hipSetDevice(rx->device);
        for (int k = 0; k < STAGES; k++) if (rx->h_stage[k]) (void)hipHostFree(rx->h_stage[k]);
        delete rx;
        return;
    }
    if (rx->ctx) {
        (void)hipSetDevice(rx->ctx->device);
        if (rx->a) (void)hipStreamSynchronize(rx->a);
    }
    if (rx->ses) dabgpu_frame_session_destroy(rx->ses);          // (synchronises stream B)
    for (int k = 0; k < STAGES; k++) {
        if (rx->h_stage[k]) (void)hipHostFree(rx->h_stage[k]);
        if (rx->stage_free[k]) (void)hipEventDestroy(rx->stage_free[k]);
    }
    if (rx->d_prs) (void)hipFree(rx->d_prs);
    if (rx->d_iq) (void)hipFree(rx->d_iq);
    if (rx->d_rec) (void)hipFree(rx->d_rec);
    if (rx->d_corr) (void)hipFree(rx->d_corr);
    if (rx->d_fft) (void)hipFree(rx->d_fft);
    if (rx->d_dq) (void)hipFree(rx->d_dq);
    if (rx->h_rec) (void)hipHostFree(rx->h_rec);
    if (rx->sync_done) (void)hipEventDestroy(rx->sync_done);
    if (rx->a) (void)hipStreamDestroy(rx->a);
    if (rx->ctx) dabgpu_destroy(rx->ctx);
    delete rx;
}

extern "C" int dabgpu_receiver_create(dabgpu_receiver** out, int device, int mode, const float* h_prs, const int* h_mapper) {
    if (!out) { dabgpu_set_error("receiver_create: null argument"); return DABGPU_ERR_INVALID_ARG; }
    *out = nullptr;
    int geom[9];
    if (dabgpu_get_ofdm_params(mode, geom) != DABGPU_OK) { dabgpu_set_error("receiver_create: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    if (mode != 1 && (h_prs || h_mapper)) { dabgpu_set_error("receiver_create: custom PRS / carrier tables are only supported in transmission mode I"); return DABGPU_ERR_INVALID_ARG; }
    dabgpu_receiver* rx = new dabgpu_receiver();
    rx->mode = mode;
    memcpy(rx->geom, geom, sizeof(geom));
    int st = dabgpu_create(&rx->ctx, device, h_prs, h_mapper);
    if (!st) st = dabgpu_frame_session_create(&rx->ses, device);
    if (!st) {
        int least = 0, greatest = 0;                                   // (numerically lower = higher priority; both 0 where priorities do not exist)
        st = dabgpu_check_hip(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        if (!st) st = dabgpu_check_hip(hipStreamCreateWithPriority(&rx->a, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority(receiver)");
    }
    const size_t n_fft = (size_t)geom[3], n_sym = (size_t)geom[0], frame_samples = (size_t)geom[6];
    // NULL symbol | frame, the frame between nb_cyclic_prefix samples before and nb_fft - nb_cyclic_prefix - 1 after the expected position
    rx->stage_cap = (size_t)geom[2] + (n_fft - (size_t)geom[4]) + frame_samples;
    for (int k = 0; k < STAGES && !st; k++) {
        st = dabgpu_check_hip(hipHostMalloc((void**)&rx->h_stage[k], rx->stage_cap * 2 * sizeof(float), hipHostMallocDefault), "hipHostMalloc(receiver stage)");
        if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&rx->stage_free[k], dabgpu_wait_event_flags(false)), "hipEventCreate(receiver)");
    }
    if (!st) st = dabgpu_check_hip(hipMalloc((void**)&rx->d_prs, n_fft * 2 * sizeof(float)), "hipMalloc(receiver)");
    if (!st) st = dabgpu_check_hip(hipMalloc((void**)&rx->d_iq, frame_samples * 2 * sizeof(float)), "hipMalloc(receiver)");
    const size_t rec_bytes = REC_HEAD + 2 * n_fft * sizeof(float);
    if (!st) st = dabgpu_check_hip(hipMalloc(&rx->d_rec, rec_bytes), "hipMalloc(receiver)");
    if (!st) st = dabgpu_check_hip(hipMemsetAsync(rx->d_rec, 0, rec_bytes, rx->a), "hipMemsetAsync(receiver)");      // (on stream A: ordered before its first synchroniser)
    if (!st) {
        rx->d_state = static_cast<dabgpu_sync_state*>(rx->d_rec);
        rx->d_small = reinterpret_cast<float*>(static_cast<unsigned char*>(rx->d_rec) + REC_SMALL);
        rx->d_imp = reinterpret_cast<float*>(static_cast<unsigned char*>(rx->d_rec) + REC_HEAD);
    }
    if (!st) st = dabgpu_check_hip(hipMalloc((void**)&rx->d_corr, n_sym * 2 * sizeof(float)), "hipMalloc(receiver)");
    if (!st) st = dabgpu_check_hip(hipHostMalloc((void**)&rx->h_rec, rec_bytes, hipHostMallocDefault), "hipHostMalloc(receiver)");
    if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&rx->sync_done, dabgpu_wait_event_flags(false)), "hipEventCreate(receiver)");
    // the result slots' page-locked soft-bit and scalar buffers now, not at the first frame that uses each slot (eight allocations of ~1 ms in the first eight frames
    // of a stream otherwise; the session frees them)
    for (int k = 0; k < dabgpu_frame_session::R && !st; k++) {
        dabgpu_frame_session::slot& sl = rx->ses->slots[k];
        st = dabgpu_check_hip(hipHostMalloc((void**)&sl.h_aux, REC_HEAD, hipHostMallocDefault), "hipHostMalloc(receiver slot)");
        if (!st) st = dabgpu_check_hip(hipHostMalloc((void**)&sl.h_bits, DABGPU_NB_FRAME_BITS, hipHostMallocDefault), "hipHostMalloc(receiver slot)");
    }
    if (st) { dabgpu_receiver_destroy(rx); return st; }
    *out = rx;
    return DABGPU_OK;
}

extern "C" int dabgpu_receiver_create_banked(dabgpu_receiver** out, int device) {
    if (!out) { dabgpu_set_error("receiver_create_banked: null argument"); return DABGPU_ERR_INVALID_ARG; }
    *out = nullptr;
    int geom[9];
    (void)dabgpu_get_ofdm_params(1, geom);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); dabgpu_set_error("hipGetDeviceCount found no device"); return DABGPU_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { dabgpu_set_error("device %d out of range (%d devices)", device, n); return DABGPU_ERR_INVALID_ARG; }
    int st = dabgpu_check_hip(hipSetDevice(device), "hipSetDevice");
    if (st) return st;
    dabgpu_receiver* rx = new dabgpu_receiver();
    rx->mode = 1;
    rx->device = device;
    memcpy(rx->geom, geom, sizeof(geom));
    const size_t n_fft = (size_t)geom[3], frame_samples = (size_t)geom[6];
    rx->stage_cap = (size_t)geom[2] + (n_fft - (size_t)geom[4]) + frame_samples;
    for (int k = 0; k < STAGES && !st; k++)
        st = dabgpu_check_hip(hipHostMalloc((void**)&rx->h_stage[k], rx->stage_cap * 2 * sizeof(float), hipHostMallocDefault), "hipHostMalloc(receiver stage)");
    if (!st) st = dabgpu_rx_bank_join(device, rx->h_stage, &rx->member);
    if (st) {
        for (int k = 0; k < STAGES; k++) if (rx->h_stage[k]) (void)hipHostFree(rx->h_stage[k]);
        delete rx;
        return st;
    }
    *out = rx;
    return DABGPU_OK;
}

extern "C" dabgpu_frame_session* dabgpu_receiver_session(dabgpu_receiver* rx) { return !rx ? nullptr : (rx->member ? dabgpu_rx_bank_session(rx->member) : rx->ses); }

extern "C" int dabgpu_receiver_set_subchannels(dabgpu_receiver* rx, const dabgpu_subchannel* subs, int n, int decode_fic) {
    if (!rx) { dabgpu_set_error("receiver_set_subchannels: null receiver"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->mode != 1 && (n > 0 || decode_fic)) { dabgpu_set_error("receiver_set_subchannels: the DAB layer above the soft bits exists for transmission mode I only"); return DABGPU_ERR_UNSUPPORTED; }
    if (rx->member) return dabgpu_rx_bank_set_subchannels(rx->member, subs, n, decode_fic);
    const int st = dabgpu_frame_session_set_subchannels(rx->ses, subs, n);
    if (st) return st;
    rx->decode_fic.store(decode_fic ? 1 : 0);
    return DABGPU_OK;
}

extern "C" int dabgpu_receiver_stage(dabgpu_receiver* rx, float** h_stage, size_t* capacity_samples) {
    if (!rx || !h_stage) { dabgpu_set_error("receiver_stage: null argument"); return DABGPU_ERR_INVALID_ARG; }
    *h_stage = rx->h_stage[rx->cur];
    if (capacity_samples) *capacity_samples = rx->stage_cap;
    return DABGPU_OK;
}

extern "C" int dabgpu_receiver_reset(dabgpu_receiver* rx) {
    if (!rx) { dabgpu_set_error("receiver_reset: null receiver"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) return dabgpu_rx_bank_reset(rx->member);
    DABGPU_BIND(rx->ctx);
    // :277-289 coarse = fine = 0, no coarse offset found -- behind everything already enqueued (a frame in flight keeps its offsets)
    return dabgpu_check_hip(hipMemsetAsync(rx->d_state, 0, sizeof(dabgpu_sync_state), rx->a), "hipMemsetAsync(receiver state)");
}

extern "C" int dabgpu_receiver_submit_sync(dabgpu_receiver* rx, const dabgpu_sync_cfg* cfg, size_t prs_sample) {
    if (!rx || !cfg) { dabgpu_set_error("receiver_submit_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    const size_t n_fft = (size_t)rx->geom[3];
    if (prs_sample + n_fft > rx->stage_cap) { dabgpu_set_error("receiver_submit_sync: the PRS slot lies outside the staging buffer"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) return dabgpu_rx_bank_post_sync(rx->member, cfg, rx->cur, prs_sample);
    if (rx->sync_pending) { dabgpu_set_error("receiver_submit_sync: the previous record has not been collected (dabgpu_receiver_wait_sync)"); return DABGPU_ERR_INVALID_ARG; }
    dabgpu_ctx* c = rx->ctx;
    DABGPU_BIND(c);
    hipStream_t a = rx->a;
    int st;
    CK(hipMemcpyAsync(rx->d_prs, rx->h_stage[rx->cur] + 2 * prs_sample, n_fft * 2 * sizeof(float), hipMemcpyHostToDevice, a));
    rx->sync_coarse = cfg->is_coarse_freq_correction != 0;
    if ((st = dabgpu_ofdm_sync_mode(c, rx->mode, rx->d_prs, 1, n_fft, cfg, rx->d_state, rx->d_imp, rx->sync_coarse ? rx->d_imp + n_fft : nullptr, a))) return st;
    // (two copies on purpose: head + responses in one is 48 bytes over 16 KiB, and a device-to-host copy above 16 KiB takes a slower path in the
    //  runtime -- +18 us until the record is on the host, tools/exp/ab_mirror.sh)
    CK(hipMemcpyAsync(rx->h_rec, rx->d_rec, REC_HEAD, hipMemcpyDeviceToHost, a));
    CK(hipMemcpyAsync(rx->h_rec + REC_HEAD, rx->d_imp, (rx->sync_coarse ? 2 : 1) * n_fft * sizeof(float), hipMemcpyDeviceToHost, a));
    CK(hipEventRecord(rx->sync_done, a));
    rx->sync_pending = true;
    return DABGPU_OK;
}

extern "C" int dabgpu_receiver_wait_sync(dabgpu_receiver* rx, dabgpu_sync_state* out, float* h_impulse, float* h_freq_response) {
    if (!rx || !out) { dabgpu_set_error("receiver_wait_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) return dabgpu_rx_bank_wait_sync(rx->member, out, h_impulse, h_freq_response);
    if (!rx->sync_pending) { dabgpu_set_error("receiver_wait_sync: no synchronisation was submitted"); return DABGPU_ERR_NOT_READY; }
    DABGPU_BIND(rx->ctx);
    int st;
    CK(hipEventSynchronize(rx->sync_done));
    rx->sync_pending = false;
    *out = *reinterpret_cast<const dabgpu_sync_state*>(rx->h_rec);
    const float* h_imp = reinterpret_cast<const float*>(rx->h_rec + REC_HEAD);
    const size_t n_fft = (size_t)rx->geom[3];
    if (h_impulse) memcpy(h_impulse, h_imp, n_fft * sizeof(float));
    if (h_freq_response && rx->sync_coarse) memcpy(h_freq_response, h_imp + n_fft, n_fft * sizeof(float));
    return DABGPU_OK;
}

// Stream A's share of a frame: upload, demodulation into the session's history slot, fine-frequency update, the copies the host reads.  The slot's
// ev_ready fires when the soft bits are in the history slot, ev_copied when the host copies are made.
static int submit_demod_reserved(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, uint64_t gen, int8_t* d_bits, dabgpu_frame_session::slot* sl);

static int submit_demod(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, uint64_t* generation, dabgpu_frame_session::slot** slot_out) {
    const size_t frame_samples = (size_t)rx->geom[6];
    if (frame_sample + frame_samples > rx->stage_cap) { dabgpu_set_error("receiver_submit_frame: the frame lies outside the staging buffer"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->sync_pending) { dabgpu_set_error("receiver_submit_frame: collect the synchroniser's record first (dabgpu_receiver_wait_sync)"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(rx->ctx);
    uint64_t gen = 0;
    int8_t* d_bits = nullptr;
    dabgpu_frame_session::slot* sl = nullptr;
    int st = dabgpu_session_reserve(rx->ses, rx->a, &gen, &d_bits, &sl);
    if (st) return st;
    // a failure behind the reservation gives the generation back: the session would otherwise refuse every later commit ("not the one reserved")
    // and the receiver were wedged for good
    if ((st = submit_demod_reserved(rx, frame_sample, beta, want_views, gen, d_bits, sl))) { dabgpu_session_unreserve(rx->ses, gen); return st; }
    if (generation) *generation = gen;
    if (slot_out) *slot_out = sl;
    return DABGPU_OK;
}

static int submit_demod_reserved(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, uint64_t /*gen*/, int8_t* d_bits, dabgpu_frame_session::slot* sl) {
    const size_t frame_samples = (size_t)rx->geom[6], n_fft = (size_t)rx->geom[3], n_sym = (size_t)rx->geom[0], frame_bits = (size_t)rx->geom[8];
    dabgpu_ctx* c = rx->ctx;
    DABGPU_BIND(c);
    hipStream_t a = rx->a;
    int st;
    const size_t fft_bytes = (n_sym + 1) * n_fft * 2 * sizeof(float), dq_bytes = (n_sym - 1) * (size_t)rx->geom[5] * 2 * sizeof(float);
    const bool want_dq = want_views && rx->mode == 1;
    if (want_views) {
        if (!rx->d_fft) CK(hipMalloc((void**)&rx->d_fft, fft_bytes));
        if (want_dq && !rx->d_dq) CK(hipMalloc((void**)&rx->d_dq, dq_bytes));
        if (sl->h_fft_cap < fft_bytes) {
            if (sl->h_fft) (void)hipHostFree(sl->h_fft);
            sl->h_fft = nullptr; sl->h_fft_cap = 0;
            CK(hipHostMalloc((void**)&sl->h_fft, fft_bytes, hipHostMallocDefault));
            sl->h_fft_cap = fft_bytes;
        }
        if (want_dq && sl->h_dq_cap < dq_bytes) {
            if (sl->h_dq) (void)hipHostFree(sl->h_dq);
            sl->h_dq = nullptr; sl->h_dq_cap = 0;
            CK(hipHostMalloc((void**)&sl->h_dq, dq_bytes, hipHostMallocDefault));
            sl->h_dq_cap = dq_bytes;
        }
    }
    if (!sl->h_aux) CK(hipHostMalloc((void**)&sl->h_aux, REC_HEAD, hipHostMallocDefault));
    if (!sl->h_bits) CK(hipHostMalloc((void**)&sl->h_bits, DABGPU_NB_FRAME_BITS, hipHostMallocDefault));
    CK(hipMemcpyAsync(rx->d_iq, rx->h_stage[rx->cur] + 2 * frame_sample, frame_samples * 2 * sizeof(float), hipMemcpyHostToDevice, a));
    CK(hipEventRecord(rx->stage_free[rx->cur], a));
    rx->stage_pending[rx->cur] = true;
    rx_net_freq_kernel<<<1, 64, 0, a>>>(rx->d_state, rx->d_small);
    CK(hipGetLastError());
    float* d_fine = &rx->d_state->freq_fine;
    if (rx->mode == 1) {
        if ((st = dabgpu_ofdm_demod_frames(c, rx->d_iq, 1, rx->d_small, d_bits, rx->d_corr, want_views ? rx->d_fft : nullptr, want_dq ? rx->d_dq : nullptr, 0, 0, a))) return st;
        if ((st = dabgpu_ofdm_phase_update(c, rx->d_corr, 1, beta, rx->d_small + 2, d_fine, a))) return st;
    } else {
        if ((st = dabgpu_ofdm_demod_frames_mode(c, rx->mode, rx->d_iq, 1, rx->d_small, d_bits, rx->d_corr, want_views ? rx->d_fft : nullptr, 0, a))) return st;
        if ((st = dabgpu_ofdm_phase_update_mode(c, rx->mode, rx->d_corr, 1, beta, rx->d_small + 2, d_fine, a))) return st;
    }
    // the decode may start; what the host reads of this frame -- soft bits, h_aux (the head of the device record: the frequency state after this frame's update, the sum of
    // the cyclic-prefix angles), the display views -- is copied on THIS stream beside it (the 230 KB of soft bits used to sit on the session's
    // stream in front of every trellis launch); the slot's done event waits for both
    CK(hipEventRecord(sl->ev_ready, a));
    CK(hipMemcpyAsync(sl->h_bits, d_bits, frame_bits, hipMemcpyDeviceToHost, a));
    CK(hipMemcpyAsync(sl->h_aux, rx->d_rec, REC_HEAD, hipMemcpyDeviceToHost, a));
    if (want_views) CK(hipMemcpyAsync(sl->h_fft, rx->d_fft, fft_bytes, hipMemcpyDeviceToHost, a));
    if (want_dq) CK(hipMemcpyAsync(sl->h_dq, rx->d_dq, dq_bytes, hipMemcpyDeviceToHost, a));
    CK(hipEventRecord(sl->ev_copied, a));
    // the next frame is assembled in the next staging buffer; its last upload (STAGES frames ago) has long finished
    rx->cur = (rx->cur + 1) % STAGES;
    if (rx->stage_pending[rx->cur]) {
        CK(hipEventSynchronize(rx->stage_free[rx->cur]));
        rx->stage_pending[rx->cur] = false;
    }
    return DABGPU_OK;
}

// banked: post the frame, move on to the next staging buffer (waiting until the bank has uploaded what it still holds)
static int submit_banked(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, int tie_rule, uint64_t* generation) {
    const size_t frame_samples = (size_t)rx->geom[6];
    if (frame_sample + frame_samples > rx->stage_cap) { dabgpu_set_error("receiver_submit_frame: the frame lies outside the staging buffer"); return DABGPU_ERR_INVALID_ARG; }
    if (dabgpu_rx_bank_sync_pending(rx->member)) { dabgpu_set_error("receiver_submit_frame: collect the synchroniser's record first (dabgpu_receiver_wait_sync)"); return DABGPU_ERR_INVALID_ARG; }
    int st = dabgpu_rx_bank_post_frame(rx->member, rx->cur, frame_sample, beta, want_views, tie_rule, generation);
    if (st) return st;
    rx->cur = (rx->cur + 1) % STAGES;
    return dabgpu_rx_bank_wait_stage(rx->member, rx->cur);
}

// one call, one thread: the decode is enqueued behind the demodulation with device-side waits (stream B waits for ev_ready, and for ev_copied before
// the slot's done event)
extern "C" int dabgpu_receiver_submit_frame(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, int tie_rule, uint64_t* generation) {
    if (!rx) { dabgpu_set_error("receiver_submit_frame: null receiver"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) return submit_banked(rx, frame_sample, beta, want_views, tie_rule, generation);
    uint64_t gen = 0;
    dabgpu_frame_session::slot* sl = nullptr;
    int st = submit_demod(rx, frame_sample, beta, want_views, &gen, &sl);
    if (st) return st;
    DABGPU_BIND(rx->ctx);
    const int fic = rx->decode_fic.load();
    bool decode;
    { std::lock_guard<std::mutex> lock(rx->ses->mu); decode = rx->mode == 1 && (fic || !rx->ses->subs.empty()); }
    if ((st = dabgpu_session_commit(rx->ses, gen, sl->ev_ready, 0, decode ? 1 : 0, fic, tie_rule, sl->ev_copied))) return st;
    if (generation) *generation = gen;
    return DABGPU_OK;
}

// two calls, possibly two threads: submit_demod enqueues stream A's share and returns; submit_decode waits ON THE HOST until the frame is demodulated
// and then enqueues the decode.  No device-side wait of one stream for another remains: with several receivers in a process two of them share a
// hardware queue, and a queue whose head waits for another queue's event holds back the other receiver's work behind it.
extern "C" int dabgpu_receiver_submit_demod(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, uint64_t* generation) {
    if (!rx) { dabgpu_set_error("receiver_submit_demod: null receiver"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) {   // a round decodes what it demodulates, with the core model the frame was posted with: dabgpu_receiver_submit_frame carries it
        dabgpu_set_error("receiver_submit_demod: a banked receiver takes dabgpu_receiver_submit_frame (one call, with the core model)");
        return DABGPU_ERR_UNSUPPORTED;
    }
    return submit_demod(rx, frame_sample, beta, want_views, generation, nullptr);
}

extern "C" int dabgpu_receiver_submit_decode(dabgpu_receiver* rx, uint64_t generation, int tie_rule) {
    if (!rx) { dabgpu_set_error("receiver_submit_decode: null receiver"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) { dabgpu_set_error("receiver_submit_decode: a banked receiver takes dabgpu_receiver_submit_frame"); return DABGPU_ERR_UNSUPPORTED; }
    dabgpu_frame_session* s = rx->ses;
    DABGPU_BIND(s->ctx);
    dabgpu_frame_session::slot* sl = &s->slots[generation % dabgpu_frame_session::R];
    int st = dabgpu_check_hip(hipEventSynchronize(sl->ev_ready), "hipEventSynchronize(receiver frame demodulated)");
    if (st) return st;
    const int fic = rx->decode_fic.load();
    bool decode;
    { std::lock_guard<std::mutex> lock(s->mu); decode = rx->mode == 1 && (fic || !s->subs.empty()); }
    return dabgpu_session_commit(s, generation, nullptr, 0, decode ? 1 : 0, fic, tie_rule, nullptr);
}


extern "C" int dabgpu_receiver_wait_frame(dabgpu_receiver* rx, uint64_t generation, dabgpu_receiver_frame* out) {
    if (!rx || !out) { dabgpu_set_error("receiver_wait_frame: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (rx->member) return dabgpu_rx_bank_wait_frame(rx->member, generation, out);
    dabgpu_frame_session* s = rx->ses;
    DABGPU_BIND(s->ctx);
    dabgpu_frame_session::slot* sl = &s->slots[generation % dabgpu_frame_session::R];
    hipEvent_t done = nullptr;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        if (sl->gen != generation) { dabgpu_set_error("receiver_wait_frame: generation %llu is gone or was never submitted", (unsigned long long)generation); return DABGPU_ERR_NOT_READY; }
        if (sl->pending) done = sl->done;
    }
    // waited for outside the session's lock: the producer thread submits the next frame meanwhile (the slot is not reused before
    // R more frames were submitted -- the caller bounds the frames in flight)
    if (done) {
        int st = dabgpu_check_hip(hipEventSynchronize(done), "hipEventSynchronize(receiver frame)");
        if (st) return st;
        std::lock_guard<std::mutex> lock(s->mu);
        if (sl->gen == generation) sl->pending = false;
    }
    // (the host copies of the producer stream: already behind `done` after dabgpu_receiver_submit_frame, on their own after _submit_decode)
    { int st = dabgpu_check_hip(hipEventSynchronize(sl->ev_copied), "hipEventSynchronize(receiver copies)"); if (st) return st; }
    out->generation = generation;
    out->bits = sl->h_bits;
    out->n_bits = (size_t)rx->geom[8];
    out->freq_fine = reinterpret_cast<const dabgpu_sync_state*>(sl->h_aux)->freq_fine;
    out->total_phase = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(sl->h_aux) + REC_SMALL)[2];
    out->fft = sl->h_fft;
    out->dqpsk = sl->h_dq;
    return DABGPU_OK;
}
