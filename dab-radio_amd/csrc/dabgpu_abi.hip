// dabgpu_abi.hip -- the C ABI declared in include/dabgpu.h (host side: contexts, constant tables,
// argument checking, launches).  No CPU compute path exists here: every compute entry point needs
// a gfx950 device and fails with DABGPU_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"

int dabgpu_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return DABGPU_OK;
    dabgpu_set_error("%s: %s", what, hipGetErrorString(e));
    return DABGPU_ERR_HIP;
}

// the capture status of the stream an entry point launches on, asked ONCE per entry point and thread (every entry point binds its context's device
// first: that is where the memo is dropped) instead of at each of the ~10 scratch look-ups of a decode call
namespace {
struct capture_memo { hipStream_t stream; bool valid, capturing; };
thread_local capture_memo tl_capture = {nullptr, false, false};
}  // namespace

unsigned dabgpu_wait_event_flags(bool bank_default_block) {
    static const int mode = [] { const char* e = getenv("DABGPU_EVENT_WAIT"); return !e ? 0 : (strcmp(e, "block") == 0 ? 1 : 2); }();
    const bool block = mode == 1 || (mode == 0 && bank_default_block);
    return hipEventDisableTiming | (block ? hipEventBlockingSync : 0u);
}

int dabgpu_bind_device(const dabgpu_ctx* c) {
    if (!c) { dabgpu_set_error("null context"); return DABGPU_ERR_INVALID_ARG; }
    tl_capture.valid = false;
    return dabgpu_check_hip(hipSetDevice(c->device), "hipSetDevice");
}

extern "C" {

int dabgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int usable = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) usable++;
    }
    return usable;
}

// ---- context ----
int dabgpu_create(dabgpu_ctx** out, int device, const float* h_prs, const int* h_mapper) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        dabgpu_set_error("hipGetDeviceCount found no device");
        return DABGPU_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { dabgpu_set_error("device %d out of range (%d devices)", device, n); return DABGPU_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    int st = dabgpu_check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (st) return st;
    if (!strstr(prop.gcnArchName, "gfx950")) {
        dabgpu_set_error("device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
        return DABGPU_ERR_NO_DEVICE;
    }
    st = dabgpu_check_hip(hipSetDevice(device), "hipSetDevice");
    if (st) return st;

    dabgpu_ctx* c = new dabgpu_ctx();
    c->device = device;
    c->prs.resize(2 * DABGPU_NB_FFT);
    c->mapper.resize(DABGPU_NB_DATA_CARRIERS);
    if (h_prs) memcpy(c->prs.data(), h_prs, sizeof(float) * 2 * DABGPU_NB_FFT); else dabgpu_get_prs_fft_ref(1, c->prs.data());
    if (h_mapper) memcpy(c->mapper.data(), h_mapper, sizeof(int) * DABGPU_NB_DATA_CARRIERS); else dabgpu_get_carrier_mapper(1, c->mapper.data());

    // inverse of the frequency de-interleaver: output position n of carrier c (mapper[n] = c)
    std::vector<uint16_t> inv(DABGPU_NB_DATA_CARRIERS, 0xFFFF);
    for (int i = 0; i < DABGPU_NB_DATA_CARRIERS; i++) {
        const int cidx = c->mapper[i];
        if (cidx < 0 || cidx >= DABGPU_NB_DATA_CARRIERS || inv[cidx] != 0xFFFF) {
            dabgpu_set_error("carrier_mapper is not a permutation of 0..1535 (entry %d = %d)", i, cidx);
            delete c;
            return DABGPU_ERR_INVALID_ARG;
        }
        inv[cidx] = (uint16_t)i;
    }
    std::vector<float> tw(2 * DABGPU_NB_FFT);
    dabgpu_get_fft_twiddles(tw.data());

#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) { dabgpu_destroy(c); return st; } } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipMalloc(&c->d_tw, sizeof(float) * tw.size()));
    CK(hipMalloc(&c->d_inv_map, sizeof(uint16_t) * inv.size()));
    CK(hipMalloc(&c->d_prs, sizeof(float) * c->prs.size()));
    CK(hipMemcpy(c->d_tw, tw.data(), sizeof(float) * tw.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_inv_map, inv.data(), sizeof(uint16_t) * inv.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_prs, c->prs.data(), sizeof(float) * c->prs.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&c->d_prs_time_ref, sizeof(float) * 2 * DABGPU_NB_FFT));
    CK(dabgpu_launch_sync_init(c->d_prs, c->d_tw, c->d_prs_time_ref, DABGPU_NB_FFT, c->stream));     // ofdm_demodulator.cpp:134-140
    CK(hipStreamSynchronize(c->stream));
#undef CK
    *out = c;
    return DABGPU_OK;
}

void dabgpu_destroy(dabgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_tw) (void)hipFree(c->d_tw);
    if (c->d_inv_map) (void)hipFree(c->d_inv_map);
    if (c->d_prs) (void)hipFree(c->d_prs);
    if (c->d_prs_time_ref) (void)hipFree(c->d_prs_time_ref);
    if (c->d_vit_tables) (void)hipFree(c->d_vit_tables);
    for (int* p : c->d_mode_mapper) if (p) (void)hipFree(p);
    for (int* p : c->d_mode_inv_map) if (p) (void)hipFree(p);
    for (float* p : c->d_mode_prs) if (p) (void)hipFree(p);
    for (float* p : c->d_mode_prs_time_ref) if (p) (void)hipFree(p);
    for (void* p : c->scratch) if (p) (void)hipFree(p);
    for (void* p : c->parked) if (p) (void)hipFree(p);
    for (auto& sl : c->stage) {
        if (sl.pending) (void)hipEventSynchronize(sl.ev);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.h) (void)hipHostFree(sl.h);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dabgpu_stage_h2d(dabgpu_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return DABGPU_OK;
    int st;
    // The ring is for small, short-lived sources (plans, lane tables, one CIF, one descriptor).  A large table (the descriptor
    // array of dabgpu_viterbi_decode_batch: tens of MB) goes through the runtime's own path: no second host copy, and no pinned slot
    // grows to the largest table ever seen.  The contract is "h_src is consumed when this returns": the runtime guarantees that for
    // PAGEABLE memory only (it stages the bytes itself before returning).  A page-locked source (hipHostMalloc, hipHostRegister --
    // dabgpu_host_pin --, a pinned torch tensor) is read by the DMA engine after the call has returned, so the copy is waited for.
    constexpr size_t STAGE_MAX = (size_t)1 << 20;
    if (bytes > STAGE_MAX) {
        hipPointerAttribute_t at;
        bool page_locked = true;                                       // unknown -> assume the worse
        const hipError_t qa = hipPointerGetAttributes(&at, h_src);
        if (qa == hipSuccess) page_locked = (at.type != hipMemoryTypeUnregistered);
        else { (void)hipGetLastError(); page_locked = (qa != hipErrorInvalidValue); }      // invalid value = a pointer HIP has never seen: pageable
        if ((st = dabgpu_check_hip(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync(large table)"))) return st;
        if (!page_locked) return DABGPU_OK;
        hipEvent_t ev = nullptr;
        if ((st = dabgpu_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate(large table)"))) return st;
        st = dabgpu_check_hip(hipEventRecord(ev, s), "hipEventRecord(large table)");
        if (!st) st = dabgpu_check_hip(hipEventSynchronize(ev), "hipEventSynchronize(large table)");
        (void)hipEventDestroy(ev);
        return st;
    }
    dabgpu_ctx::stage_slot* sl;
    {
        std::lock_guard<std::mutex> g(c->stage_mu);
        sl = &c->stage[c->stage_next++ % 8];
    }
    // per-slot lock: a thread that has to wait for its slot's last DMA holds up only the callers that wrap around to the same slot
    std::lock_guard<std::mutex> g(sl->mu);
    if (sl->pending) {
        if ((st = dabgpu_check_hip(hipEventSynchronize(sl->ev), "hipEventSynchronize(stage)"))) return st;
        sl->pending = false;
    }
    if (!sl->ev && (st = dabgpu_check_hip(hipEventCreateWithFlags(&sl->ev, hipEventDisableTiming), "hipEventCreate(stage)"))) return st;
    if (sl->bytes < bytes) {
        if (sl->h) { (void)hipHostFree(sl->h); sl->h = nullptr; sl->bytes = 0; }
        const size_t want = bytes < (size_t)65536 ? (size_t)65536 : STAGE_MAX;
        if ((st = dabgpu_check_hip(hipHostMalloc(&sl->h, want, hipHostMallocDefault), "hipHostMalloc(stage)"))) return st;
        sl->bytes = want;
    }
    memcpy(sl->h, h_src, bytes);
    if ((st = dabgpu_check_hip(hipMemcpyAsync(d_dst, sl->h, bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync(stage)"))) return st;
    if ((st = dabgpu_check_hip(hipEventRecord(sl->ev, s), "hipEventRecord(stage)"))) return st;
    sl->pending = true;
    return DABGPU_OK;
}

int dabgpu_stage_h2d_cached(dabgpu_ctx* c, int which, void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    dabgpu_ctx::table_copy& T = c->tables[which];
    std::lock_guard<std::mutex> g(c->tables_mu);
    if (T.d == d_dst && T.s == s && T.bytes.size() == bytes && memcmp(T.bytes.data(), h_src, bytes) == 0) return DABGPU_OK;
    T.d = nullptr;                                                     // (invalid until the copy below has been enqueued)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        dabgpu_set_error("this call uploads a table it has not uploaded before (first call with these sub-channels on this stream): run it once before capturing");
        return DABGPU_ERR_INVALID_ARG;
    }
    const int st = dabgpu_stage_h2d(c, d_dst, h_src, bytes, s);
    if (st) return st;
    T.bytes.assign(static_cast<const unsigned char*>(h_src), static_cast<const unsigned char*>(h_src) + bytes);
    T.d = d_dst; T.s = s;
    return DABGPU_OK;
}

// host buffers a caller hands to the *_host_sync entry points again and again (the mirror classes' frame buffers): page-locked, their
// copies run at PCIe speed instead of through the runtime's staging
int dabgpu_host_pin(void* p, size_t bytes) {
    if (!p || bytes == 0) { dabgpu_set_error("host_pin: null buffer"); return DABGPU_ERR_INVALID_ARG; }
    return dabgpu_check_hip(hipHostRegister(p, bytes, hipHostRegisterPortable), "hipHostRegister");
}
int dabgpu_host_unpin(void* p) {
    if (!p) return DABGPU_OK;
    return dabgpu_check_hip(hipHostUnregister(p), "hipHostUnregister");
}

int dabgpu_synchronize(dabgpu_ctx* c, void* stream) {
    if (!c) return DABGPU_ERR_INVALID_ARG;
    DABGPU_BIND(c);
    return dabgpu_check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
}

// ---- OFDM ----
// symbols_per_block = 0: which run length is fastest depends on the batch and on the box (a whole frame per workgroup
// is one static round of workgroups on a full chip -- fastest where all CUs run at one speed, much slower where they do not; three
// runs per frame let the dispatcher rebalance but transform two halo symbols more).  The data path never measures: it takes what
// dabgpu_ofdm_tune has recorded for the kernel variant and the batch's size bucket (nearest recorded bucket of that variant), else 25;
// small batches take shorter runs.
// what symbols_per_block = 0 resolves to; never blocks, never launches
static int demod_cached_spb(dabgpu_ctx* c, size_t n_frames, int variant) {
    if (n_frames < 512) return dabgpu_host_small_batch_spb(n_frames);
    const int want = dabgpu_host_spb_bucket(n_frames);
    DABGPU_HOST_LOCK(c);
    int best = 25, best_d = 1 << 30;
    for (const auto& e : c->spb_cache) {
        if (e.variant != variant) continue;
        const int d = e.bucket > want ? e.bucket - want : want - e.bucket;
        if (d < best_d) { best_d = d; best = e.spb; }
    }
    return best;
}

extern "C" int dabgpu_ofdm_auto_symbols_per_block(dabgpu_ctx* c, size_t n_frames) {
    if (!c) return 0;
    if (n_frames < 512) return dabgpu_host_small_batch_spb(n_frames);
    const int want = dabgpu_host_spb_bucket(n_frames);
    DABGPU_HOST_LOCK(c);
    for (auto e = c->spb_cache.rbegin(); e != c->spb_cache.rend(); ++e) if (e->bucket == want) return e->spb;      // the latest of any variant
    return 0;
}

static int ofdm_demod_any(dabgpu_ctx* c, const void* d_iq, int src, size_t n_frames, const float* d_freq, int8_t* d_bits,
                         float* d_cp_corr, float* d_fft, float* d_dqpsk, int symbols_per_block, size_t bits_frame_stride, void* stream,
                         int bits_layout = DABGPU_BITS_NATURAL, float* d_total_phase = nullptr, float* d_fine_freq = nullptr, float beta = 0.0f) {
    if (bits_layout != DABGPU_BITS_NATURAL && bits_layout != DABGPU_BITS_MSC_CLASSED) {
        dabgpu_set_error("ofdm_demod_frames: unknown bits_layout %d", bits_layout); return DABGPU_ERR_INVALID_ARG;
    }
    if (!c || !d_iq || !d_bits) { dabgpu_set_error("ofdm_demod_frames: null ctx/iq/bits"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    if (n_frames > (size_t)(1 << 24)) { dabgpu_set_error("ofdm_demod_frames: n_frames too large"); return DABGPU_ERR_INVALID_ARG; }
    if (bits_frame_stride != 0 && (bits_frame_stride < DABGPU_NB_FRAME_BITS || (bits_frame_stride & 15))) {
        dabgpu_set_error("ofdm_demod_frames: bits_frame_stride must be 0 or a multiple of 16 >= 230400"); return DABGPU_ERR_INVALID_ARG;
    }
    if (((uintptr_t)d_iq & 15) || ((uintptr_t)d_bits & 15)) { dabgpu_set_error("ofdm_demod_frames: d_iq and d_bits must be 16-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;      // NULL = the HIP default (null) stream
    float* corr = d_cp_corr;
    if (!corr) {     // the kernel always produces the correlation; park it in context scratch when unwanted
        int st = dabgpu_scratch(c, 0, n_frames * DABGPU_NB_FRAME_SYMBOLS * 2 * sizeof(float), (void**)&corr, s);
        if (st) return st;
    }
    if (symbols_per_block <= 0)
        symbols_per_block = (d_fft || d_dqpsk) ? dabgpu_host_small_batch_spb(n_frames)      /* display views: three workgroups per CU, not tuned */
                                               : demod_cached_spb(c, n_frames, dabgpu_host_spb_variant(src, bits_layout, d_total_phase || d_fine_freq));
    return dabgpu_check_hip(dabgpu_launch_ofdm_demod(d_iq, src, d_freq, d_bits, corr, d_fft, d_dqpsk, c->d_tw, c->d_inv_map,
                                                     (int)n_frames, symbols_per_block, bits_frame_stride, nullptr, nullptr, 0,
                                                     bits_layout == DABGPU_BITS_MSC_CLASSED, s, d_total_phase, d_fine_freq, beta),
                            "ofdm_demod_kernel launch");
}

int dabgpu_ofdm_demod_frames(dabgpu_ctx* c, const float* d_iq, size_t n_frames, const float* d_freq, int8_t* d_bits,
                             float* d_cp_corr, float* d_fft, float* d_dqpsk, int symbols_per_block, size_t bits_frame_stride, void* stream) {
    return ofdm_demod_any(c, d_iq, 0, n_frames, d_freq, d_bits, d_cp_corr, d_fft, d_dqpsk, symbols_per_block, bits_frame_stride, stream);
}

// capture formats the demodulator's loader dequantises itself (iq_decode.h): 0 = complex float, 1 = u8, 2 = s8, 3 = s16 little endian
static int fused_loader_of(int format) {
    switch (format) {
    case DABGPU_IQ_RAW_F32L: case DABGPU_IQ_WAV_F32: return 0;
    case DABGPU_IQ_RAW_U8: case DABGPU_IQ_WAV_PCM8: return 1;
    case DABGPU_IQ_RAW_S8: return 2;
    case DABGPU_IQ_RAW_S16L: case DABGPU_IQ_WAV_PCM16: return 3;
    default: return -1;
    }
}

int dabgpu_ofdm_demod_frames_raw(dabgpu_ctx* c, const void* d_raw, int format, size_t n_frames, const float* d_freq, int8_t* d_bits,
                                 float* d_cp_corr, float* d_fft, float* d_dqpsk, int symbols_per_block, size_t bits_frame_stride, void* stream) {
    const int src = fused_loader_of(format);
    if (src >= 0) return ofdm_demod_any(c, d_raw, src, n_frames, d_freq, d_bits, d_cp_corr, d_fft, d_dqpsk, symbols_per_block, bits_frame_stride, stream);
    // formats without a fused loader: convert into context scratch on the same stream, then demodulate
    if (!c) { dabgpu_set_error("ofdm_demod_frames_raw: null context"); return DABGPU_ERR_INVALID_ARG; }
    if (dabgpu_iq_format_sample_bytes(format) == 0) { dabgpu_set_error("ofdm_demod_frames_raw: unknown format %d", format); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    float* d_iq;
    int st = dabgpu_scratch(c, 22, n_frames * DABGPU_NB_FRAME_SAMPLES * 2 * sizeof(float), (void**)&d_iq);
    if (st) return st;
    if ((st = dabgpu_iq_convert(c, d_raw, format, n_frames * DABGPU_NB_FRAME_SAMPLES, d_iq, stream))) return st;
    return ofdm_demod_any(c, d_iq, 0, n_frames, d_freq, d_bits, d_cp_corr, d_fft, d_dqpsk, symbols_per_block, bits_frame_stride, stream);
}

int dabgpu_ofdm_demod_phase_frames(dabgpu_ctx* c, const void* d_raw, int format, size_t n_frames, const float* d_freq, int8_t* d_bits,
                                   float* d_cp_corr, int symbols_per_block, size_t bits_frame_stride, int bits_layout,
                                   float fine_freq_update_beta, float* d_total_phase, float* d_fine_freq, void* stream) {
    const int src = fused_loader_of(format);
    if (src < 0) { dabgpu_set_error("ofdm_demod_phase_frames: format %d has no fused loader (float32, u8, s8, s16 little endian do)", format); return DABGPU_ERR_INVALID_ARG; }
    return ofdm_demod_any(c, d_raw, src, n_frames, d_freq, d_bits, d_cp_corr, nullptr, nullptr, symbols_per_block, bits_frame_stride, stream, bits_layout,
                          d_total_phase, d_fine_freq, fine_freq_update_beta);
}

int dabgpu_ofdm_demod_frames_history(dabgpu_ctx* c, const void* d_raw, int format, size_t n_frames, const float* d_freq, int8_t* d_bits,
                                     float* d_cp_corr, int symbols_per_block, size_t bits_frame_stride, int bits_layout, void* stream) {
    const int src = fused_loader_of(format);
    if (src < 0) { dabgpu_set_error("ofdm_demod_frames_history: format %d has no fused loader (float32, u8, s8, s16 little endian do)", format); return DABGPU_ERR_INVALID_ARG; }
    return ofdm_demod_any(c, d_raw, src, n_frames, d_freq, d_bits, d_cp_corr, nullptr, nullptr, symbols_per_block, bits_frame_stride, stream, bits_layout);
}

// Explicit calibration of symbols_per_block = 0 (the data path never measures).  Blocks the calling thread: untimed launches until
// 40 ms of kernels have run (the clock of an idle GPU settles over tens of milliseconds, profiles/r01/ab_notes.md; at most 12 rounds),
// then three timed rounds that visit the candidates in turn, two launches each, summed per candidate -- a candidate is never judged by one
// moment of the clock.  Every candidate is timed as it will run: with_phase_tail adds the tail (inside the kernel at 75, a second launch
// at 25 / 38) on context scratch, not on the caller's fine-frequency state.  d_bits receives valid soft bits (identical for every run length).
int dabgpu_ofdm_tune(dabgpu_ctx* c, const void* d_raw, int format, size_t n_frames, int8_t* d_bits, size_t bits_frame_stride, int bits_layout,
                     int with_phase_tail, void* stream, int* chosen) {
    const int src = fused_loader_of(format);
    if (chosen) *chosen = 0;
    if (src < 0) { dabgpu_set_error("ofdm_tune: format %d has no fused loader (float32, u8, s8, s16 little endian do)", format); return DABGPU_ERR_INVALID_ARG; }
    if (bits_layout != DABGPU_BITS_NATURAL && bits_layout != DABGPU_BITS_MSC_CLASSED) { dabgpu_set_error("ofdm_tune: unknown bits_layout %d", bits_layout); return DABGPU_ERR_INVALID_ARG; }
    if (!c || !d_raw || !d_bits) { dabgpu_set_error("ofdm_tune: null ctx/iq/bits"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0 || n_frames > (size_t)(1 << 24)) { dabgpu_set_error("ofdm_tune: n_frames out of range"); return DABGPU_ERR_INVALID_ARG; }
    if (bits_frame_stride != 0 && (bits_frame_stride < DABGPU_NB_FRAME_BITS || (bits_frame_stride & 15))) {
        dabgpu_set_error("ofdm_tune: bits_frame_stride must be 0 or a multiple of 16 >= 230400"); return DABGPU_ERR_INVALID_ARG;
    }
    if (((uintptr_t)d_raw & 15) || ((uintptr_t)d_bits & 15)) { dabgpu_set_error("ofdm_tune: d_raw and d_bits must be 16-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames < 512) { if (chosen) *chosen = dabgpu_host_small_batch_spb(n_frames); return DABGPU_OK; }      // nothing to measure: the small-batch rule
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError(); dabgpu_set_error("ofdm_tune: the stream is capturing"); return DABGPU_ERR_INVALID_ARG;
    }
    float *corr, *tail = nullptr;
    int st = dabgpu_scratch(c, 0, n_frames * DABGPU_NB_FRAME_SYMBOLS * 2 * sizeof(float), (void**)&corr);
    if (st) return st;
    if (with_phase_tail) {
        if ((st = dabgpu_scratch(c, 23, 2 * n_frames * sizeof(float), (void**)&tail))) return st;
        if ((st = dabgpu_check_hip(hipMemsetAsync(tail, 0, 2 * n_frames * sizeof(float), s), "hipMemsetAsync(tune)"))) return st;
    }
    auto launch = [&](int spb) {
        return dabgpu_launch_ofdm_demod(d_raw, src, nullptr, d_bits, corr, nullptr, nullptr, c->d_tw, c->d_inv_map, (int)n_frames, spb, bits_frame_stride,
                                        nullptr, nullptr, 0, bits_layout == DABGPU_BITS_MSC_CLASSED, s, tail, tail ? tail + n_frames : nullptr, 0.9f);
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if ((st = dabgpu_check_hip(hipEventCreate(&e0), "hipEventCreate(tune)"))) return st;
    if ((st = dabgpu_check_hip(hipEventCreate(&e1), "hipEventCreate(tune)"))) { (void)hipEventDestroy(e0); return st; }
    static const int cand[3] = {25, 38, 75};
    float sum_ms[3] = {0.0f, 0.0f, 0.0f}, warm_ms = 0.0f;
    hipError_t err = hipSuccess;
    for (int pass = 0, timed = 0; timed < 3 && pass < 16 && err == hipSuccess; pass++) {
        const bool counts = warm_ms >= 40.0f || pass >= 12;
        for (int k = 0; k < 3 && err == hipSuccess; k++) {
            err = hipEventRecord(e0, s);
            for (int r = 0; r < 2 && err == hipSuccess; r++) err = launch(cand[k]);
            float ms = 0.0f;
            if (err == hipSuccess) err = hipEventRecord(e1, s);
            if (err == hipSuccess) err = hipEventSynchronize(e1);
            if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
            if (counts) sum_ms[k] += ms; else warm_ms += ms;
        }
        timed += counts;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return dabgpu_check_hip(err, "ofdm_tune");
    int best = 0;
    for (int k = 1; k < 3; k++) if (sum_ms[k] < sum_ms[best]) best = k;
    const int bucket = dabgpu_host_spb_bucket(n_frames), variant = dabgpu_host_spb_variant(src, bits_layout, with_phase_tail != 0);
    {
        DABGPU_HOST_LOCK(c);
        bool found = false;
        for (auto& e : c->spb_cache) if (e.bucket == bucket && e.variant == variant) { e.spb = cand[best]; found = true; }
        if (!found) c->spb_cache.push_back({bucket, variant, cand[best]});
    }
    if (chosen) *chosen = cand[best];
    return DABGPU_OK;
}

int dabgpu_ofdm_tuned_symbols_per_block(dabgpu_ctx* c, int format, size_t n_frames, int bits_layout, int with_phase_tail) {
    const int src = fused_loader_of(format);
    if (!c || src < 0 || n_frames == 0) return 0;
    return demod_cached_spb(c, n_frames, dabgpu_host_spb_variant(src, bits_layout, with_phase_tail != 0));
}

// One steady-state frame of n receivers: PRS synchronisation, then the demodulation it positions and corrects, then the fine-frequency
// update -- three dependent steps of OFDM_Demod per frame (ofdm_demodulator.cpp:360-548, :650-766, :606-618) with the records of the
// first read by the second and third on the device.
int dabgpu_ofdm_sync_demod_frames(dabgpu_ctx* c, const float* d_iq, size_t n_streams, size_t stream_stride_samples, size_t prs_offset_samples,
                                  const dabgpu_sync_cfg* cfg, dabgpu_sync_state* d_states, int8_t* d_bits, float* d_cp_corr, int symbols_per_block,
                                  size_t bits_frame_stride, int bits_layout, float* d_total_phase, void* stream) {
    if (!c || !d_iq || !cfg || !d_states || !d_bits) { dabgpu_set_error("ofdm_sync_demod_frames: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (bits_layout != DABGPU_BITS_NATURAL && bits_layout != DABGPU_BITS_MSC_CLASSED) { dabgpu_set_error("ofdm_sync_demod_frames: unknown bits_layout %d", bits_layout); return DABGPU_ERR_INVALID_ARG; }
    if (n_streams == 0) return DABGPU_OK;
    if (n_streams > (size_t)(1 << 24)) { dabgpu_set_error("ofdm_sync_demod_frames: n_streams too large"); return DABGPU_ERR_INVALID_ARG; }
    // the impulse peak lies in [0, 2048): the frame starts between 504 samples before and 1543 after the expected position
    if (prs_offset_samples < DABGPU_NB_CYCLIC_PREFIX || prs_offset_samples > (size_t)(1 << 30) ||
        stream_stride_samples < prs_offset_samples + (DABGPU_NB_FFT - DABGPU_NB_CYCLIC_PREFIX) + (size_t)DABGPU_NB_FRAME_SYMBOLS * DABGPU_NB_SYMBOL_PERIOD) {
        dabgpu_set_error("ofdm_sync_demod_frames: needs prs_offset_samples >= 504 and stream_stride_samples >= prs_offset_samples + 1544 + 76 * 2552");
        return DABGPU_ERR_INVALID_ARG;
    }
    if (bits_frame_stride != 0 && (bits_frame_stride < DABGPU_NB_FRAME_BITS || (bits_frame_stride & 15))) {
        dabgpu_set_error("ofdm_sync_demod_frames: bits_frame_stride must be 0 or a multiple of 16 >= 230400"); return DABGPU_ERR_INVALID_ARG;
    }
    if (((uintptr_t)d_iq & 7) || ((uintptr_t)d_bits & 15)) { dabgpu_set_error("ofdm_sync_demod_frames: d_iq must be 8-byte, d_bits 16-byte aligned"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    float* corr = d_cp_corr;
    int st;
    if (!corr && (st = dabgpu_scratch(c, 0, n_streams * DABGPU_NB_FRAME_SYMBOLS * 2 * sizeof(float), (void**)&corr, s))) return st;
    if ((st = dabgpu_check_hip(dabgpu_launch_sync(d_iq + 2 * prs_offset_samples, stream_stride_samples, (int)n_streams, cfg, d_states, nullptr, nullptr,
                                                 c->d_tw, c->d_prs, c->d_prs_time_ref, nullptr, 1, s), "ofdm_sync_kernel launch"))) return st;
    if (symbols_per_block <= 0) symbols_per_block = demod_cached_spb(c, n_streams, dabgpu_host_spb_variant(0, bits_layout, true));
    return dabgpu_check_hip(dabgpu_launch_ofdm_demod(d_iq, 0, nullptr, d_bits, corr, nullptr, nullptr, c->d_tw, c->d_inv_map, (int)n_streams, symbols_per_block,
                                                     bits_frame_stride, nullptr, nullptr, 0, bits_layout == DABGPU_BITS_MSC_CLASSED, s, d_total_phase, nullptr,
                                                     cfg->fine_freq_update_beta, nullptr, stream_stride_samples, d_states, (int)prs_offset_samples),
                            "ofdm_demod_kernel launch (synchronised frames)");
}

int dabgpu_ofdm_phase_update_mode(dabgpu_ctx* c, int mode, const float* d_cp_corr, size_t n_frames, float beta, float* d_total_phase,
                                  float* d_fine_freq, void* stream) {
    int geom[9];
    if (!c || !d_cp_corr) { dabgpu_set_error("ofdm_phase_update_mode: null ctx/corr"); return DABGPU_ERR_INVALID_ARG; }
    if (dabgpu_get_ofdm_params(mode, geom)) return DABGPU_ERR_INVALID_ARG;
    if (n_frames == 0) return DABGPU_OK;
    DABGPU_BIND(c);
    return dabgpu_check_hip(dabgpu_launch_ofdm_phase(d_cp_corr, (int)n_frames, beta, d_total_phase, d_fine_freq, 1, nullptr, geom[0], geom[3],
                                                     (hipStream_t)stream), "ofdm_phase_kernel launch");
}

int dabgpu_ofdm_phase_update(dabgpu_ctx* c, const float* d_cp_corr, size_t n_frames, float beta, float* d_total_phase,
                             float* d_fine_freq, void* stream) {
    if (!c || !d_cp_corr) { dabgpu_set_error("ofdm_phase_update: null ctx/corr"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    return dabgpu_check_hip(dabgpu_launch_ofdm_phase(d_cp_corr, (int)n_frames, beta, d_total_phase, d_fine_freq, 1, nullptr, DABGPU_NB_FRAME_SYMBOLS,
                                                     DABGPU_NB_FFT, s),
                            "ofdm_phase_kernel launch");
}

int dabgpu_ofdm_demod_frames_host_sync(dabgpu_ctx* c, const float* h_iq, size_t n_frames, const float* h_freq,
                                       int8_t* h_bits, float* h_total_phase, float* h_fft) {
    if (!c || !h_iq || !h_bits) { dabgpu_set_error("ofdm_demod_frames_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    int st;
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(c);
    const size_t iq_bytes = n_frames * DABGPU_NB_FRAME_SAMPLES * 2 * sizeof(float);
    const size_t bits_bytes = n_frames * DABGPU_NB_FRAME_BITS;
    const size_t fft_bytes = n_frames * 77 * DABGPU_NB_FFT * 2 * sizeof(float);
    float *d_iq, *d_freq, *d_corr, *d_total, *d_fft = nullptr; int8_t* d_bits;
    if ((st = dabgpu_scratch(c, 1, iq_bytes, (void**)&d_iq))) return st;
    if ((st = dabgpu_scratch(c, 2, bits_bytes, (void**)&d_bits))) return st;
    if ((st = dabgpu_scratch(c, 3, n_frames * sizeof(float), (void**)&d_freq))) return st;
    if ((st = dabgpu_scratch(c, 4, n_frames * DABGPU_NB_FRAME_SYMBOLS * 2 * sizeof(float), (void**)&d_corr))) return st;
    if ((st = dabgpu_scratch(c, 5, n_frames * sizeof(float), (void**)&d_total))) return st;
    if (h_fft && (st = dabgpu_scratch(c, 6, fft_bytes, (void**)&d_fft))) return st;
    hipStream_t s = c->stream;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    CK(hipMemcpyAsync(d_iq, h_iq, iq_bytes, hipMemcpyHostToDevice, s));
    if (h_freq) CK(hipMemcpyAsync(d_freq, h_freq, n_frames * sizeof(float), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_ofdm_demod_frames(c, d_iq, n_frames, h_freq ? d_freq : nullptr, d_bits, d_corr, d_fft, nullptr, 0, 0, s))) return st;
    if ((st = dabgpu_ofdm_phase_update(c, d_corr, n_frames, 0.0f, d_total, nullptr, s))) return st;
    CK(hipMemcpyAsync(h_bits, d_bits, bits_bytes, hipMemcpyDeviceToHost, s));
    if (h_total_phase) CK(hipMemcpyAsync(h_total_phase, d_total, n_frames * sizeof(float), hipMemcpyDeviceToHost, s));
    if (h_fft) CK(hipMemcpyAsync(h_fft, d_fft, fft_bytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
#undef CK
    return DABGPU_OK;
}

int dabgpu_ofdm_demod_stream_frame_sync(dabgpu_ctx* c, const float* h_iq, float freq_coarse, float* h_freq_fine, float beta,
                                        int8_t* h_bits, float* h_total_phase, float* h_fft, float* h_dqpsk) {
    if (!c || !h_iq || !h_bits || !h_freq_fine) { dabgpu_set_error("ofdm_demod_stream_frame_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    int st;
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(c);
    const size_t iq_bytes = (size_t)DABGPU_NB_FRAME_SAMPLES * 2 * sizeof(float);
    const size_t fft_bytes = (size_t)77 * DABGPU_NB_FFT * 2 * sizeof(float);
    const size_t dq_bytes = (size_t)75 * DABGPU_NB_DATA_CARRIERS * 2 * sizeof(float);
    float *d_iq, *d_small, *d_corr, *d_fft = nullptr, *d_dq = nullptr; int8_t* d_bits;
    if ((st = dabgpu_scratch(c, 1, iq_bytes, (void**)&d_iq))) return st;
    if ((st = dabgpu_scratch(c, 2, DABGPU_NB_FRAME_BITS, (void**)&d_bits))) return st;
    if ((st = dabgpu_scratch(c, 3, 4 * sizeof(float), (void**)&d_small))) return st;       // [0] net freq, [1] fine, [2] total phase
    if ((st = dabgpu_scratch(c, 4, DABGPU_NB_FRAME_SYMBOLS * 2 * sizeof(float), (void**)&d_corr))) return st;
    if (h_fft && (st = dabgpu_scratch(c, 6, fft_bytes, (void**)&d_fft))) return st;
    if (h_dqpsk && (st = dabgpu_scratch(c, 13, dq_bytes, (void**)&d_dq))) return st;
    hipStream_t s = c->stream;
    const float h_small[2] = { freq_coarse + *h_freq_fine, *h_freq_fine };               // :672 net offset of this frame
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    CK(hipMemcpyAsync(d_iq, h_iq, iq_bytes, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_small, h_small, sizeof(h_small), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_ofdm_demod_frames(c, d_iq, 1, d_small, d_bits, d_corr, d_fft, d_dq, 0, 0, s))) return st;
    if ((st = dabgpu_ofdm_phase_update(c, d_corr, 1, beta, d_small + 2, d_small + 1, s))) return st;
    CK(hipMemcpyAsync(h_bits, d_bits, DABGPU_NB_FRAME_BITS, hipMemcpyDeviceToHost, s));
    float back[2];
    CK(hipMemcpyAsync(back, d_small + 1, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (h_fft) CK(hipMemcpyAsync(h_fft, d_fft, fft_bytes, hipMemcpyDeviceToHost, s));
    if (h_dqpsk) CK(hipMemcpyAsync(h_dqpsk, d_dq, dq_bytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
#undef CK
    *h_freq_fine = back[0];
    if (h_total_phase) *h_total_phase = back[1];
    return DABGPU_OK;
}

// ---- sync ----
int dabgpu_ofdm_sync(dabgpu_ctx* c, const float* d_prs_syms, size_t n_streams, size_t stride_samples, const dabgpu_sync_cfg* cfg,
                     dabgpu_sync_state* d_states, float* d_impulse, float* d_freq, void* stream) {
    if (!c || !d_prs_syms || !cfg || !d_states) { dabgpu_set_error("ofdm_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_streams == 0) return DABGPU_OK;
    if (n_streams > (size_t)(1 << 24)) { dabgpu_set_error("ofdm_sync: n_streams too large"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    return dabgpu_check_hip(dabgpu_launch_sync(d_prs_syms, stride_samples, (int)n_streams, cfg, d_states, d_impulse, d_freq,
                                               c->d_tw, c->d_prs, c->d_prs_time_ref, nullptr, 1, (hipStream_t)stream), "ofdm_sync_kernel launch");
}

int dabgpu_ofdm_sync_mode(dabgpu_ctx* c, int mode, const float* d_prs_syms, size_t n_streams, size_t stride_samples, const dabgpu_sync_cfg* cfg,
                          dabgpu_sync_state* d_states, float* d_impulse, float* d_freq, void* stream) {
    if (!c || !d_prs_syms || !cfg || !d_states) { dabgpu_set_error("ofdm_sync_mode: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_streams == 0) return DABGPU_OK;
    if (n_streams > (size_t)(1 << 24)) { dabgpu_set_error("ofdm_sync_mode: n_streams too large"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    const float *d_prs, *d_ref;
    int st = dabgpu_mode_sync_tables(c, mode, &d_prs, &d_ref);
    if (st) return st;
    return dabgpu_check_hip(dabgpu_launch_sync(d_prs_syms, stride_samples, (int)n_streams, cfg, d_states, d_impulse, d_freq,
                                               c->d_tw, d_prs, d_ref, nullptr, mode, (hipStream_t)stream), "ofdm_sync_kernel launch");
}

int dabgpu_ofdm_sync_host_sync(dabgpu_ctx* c, const float* h_prs_sym, const dabgpu_sync_cfg* cfg, dabgpu_sync_state* h_state,
                               float* h_impulse, float* h_freq) {
    if (!c || !h_prs_sym || !cfg || !h_state) { dabgpu_set_error("ofdm_sync_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(c);
    int st;
    float *d_sym, *d_imp, *d_frq; dabgpu_sync_state* d_st;
    if ((st = dabgpu_scratch(c, 7, sizeof(float) * 2 * DABGPU_NB_FFT, (void**)&d_sym))) return st;
    if ((st = dabgpu_scratch(c, 8, sizeof(dabgpu_sync_state), (void**)&d_st))) return st;
    if ((st = dabgpu_scratch(c, 9, sizeof(float) * 2 * DABGPU_NB_FFT, (void**)&d_imp))) return st;
    d_frq = d_imp + DABGPU_NB_FFT;
    hipStream_t s = c->stream;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    CK(hipMemcpyAsync(d_sym, h_prs_sym, sizeof(float) * 2 * DABGPU_NB_FFT, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_st, h_state, sizeof(dabgpu_sync_state), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_ofdm_sync(c, d_sym, 1, DABGPU_NB_FFT, cfg, d_st, d_imp, d_frq, s))) return st;
    CK(hipMemcpyAsync(h_state, d_st, sizeof(dabgpu_sync_state), hipMemcpyDeviceToHost, s));
    if (h_impulse) CK(hipMemcpyAsync(h_impulse, d_imp, sizeof(float) * DABGPU_NB_FFT, hipMemcpyDeviceToHost, s));
    if (h_freq) CK(hipMemcpyAsync(h_freq, d_frq, sizeof(float) * DABGPU_NB_FFT, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
#undef CK
    return DABGPU_OK;
}

}  // extern "C"

// PRS spectrum and coarse-sync reference of modes II-IV on the device, built on first use (ofdm_demodulator.cpp:128-140)
int dabgpu_mode_sync_tables(dabgpu_ctx* c, int mode, const float** d_prs, const float** d_prs_time_ref) {
    int geom[9];
    if (dabgpu_get_ofdm_params(mode, geom)) return DABGPU_ERR_INVALID_ARG;
    if (mode == 1) { *d_prs = c->d_prs; *d_prs_time_ref = c->d_prs_time_ref; return DABGPU_OK; }
    if (!c->d_mode_prs[mode]) {
        const size_t bytes = sizeof(float) * 2 * (size_t)geom[3];
        std::vector<float> prs(2 * (size_t)geom[3]);
        int st = dabgpu_get_prs_fft_ref(mode, prs.data());
        if (st) return st;
        DABGPU_BIND(c);
        float *dp = nullptr, *dr = nullptr;
        if ((st = dabgpu_check_hip(hipMalloc(&dp, bytes), "hipMalloc(mode prs)"))) return st;
        if ((st = dabgpu_check_hip(hipMalloc(&dr, bytes), "hipMalloc(mode prs ref)"))) { (void)hipFree(dp); return st; }
        if ((st = dabgpu_check_hip(hipMemcpy(dp, prs.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy(mode prs)")) ||
            (st = dabgpu_check_hip(dabgpu_launch_sync_init(dp, c->d_tw, dr, geom[3], c->stream), "sync_init_kernel launch")) ||
            (st = dabgpu_check_hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize"))) { (void)hipFree(dp); (void)hipFree(dr); return st; }
        c->d_mode_prs[mode] = dp; c->d_mode_prs_time_ref[mode] = dr;
    }
    *d_prs = c->d_mode_prs[mode]; *d_prs_time_ref = c->d_mode_prs_time_ref[mode];
    return DABGPU_OK;
}

// grow-only scratch slots owned by the context (freed in dabgpu_destroy).
// HIP graphs: a captured graph holds the ADDRESSES of the slots it was captured with.  `user` = the stream of a capturable entry point:
//   * growth needed while `user` is capturing -> DABGPU_ERR_INVALID_ARG (hipMalloc would invalidate the capture with an obscure error);
//   * once a call of this context has run under capture, a slot that grows later (a larger eager call) PARKS its old buffer until
//     dabgpu_destroy instead of freeing it: a replay of the earlier graph then still reads and writes memory the context owns.
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out, hipStream_t user, bool user_given) {
    if ((size_t)slot >= c->scratch.size()) { c->scratch.resize(slot + 1, nullptr); c->scratch_bytes.resize(slot + 1, 0); }
    bool capturing = false;
    if (user_given) {
        if (!tl_capture.valid || tl_capture.stream != user) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool is = hipStreamIsCapturing(user, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
            if (!is) (void)hipGetLastError();
            tl_capture = {user, true, is};
        }
        if (tl_capture.capturing) { capturing = true; c->captured_once = true; }
    }
    if (c->scratch_bytes[slot] < bytes) {
        if (capturing) {
            dabgpu_set_error("this call needs more device scratch than the context holds (slot %d: %zu -> %zu bytes): run it once with these shapes before capturing", slot,
                             c->scratch_bytes[slot], bytes);
            return DABGPU_ERR_INVALID_ARG;
        }
        if (c->scratch[slot]) {
            if (c->captured_once) c->parked.push_back(c->scratch[slot]);
            else { (void)hipStreamSynchronize(c->stream); if (user_given && user != c->stream) (void)hipStreamSynchronize(user); (void)hipFree(c->scratch[slot]); }
            c->scratch[slot] = nullptr; c->scratch_bytes[slot] = 0;
        }
        int st = dabgpu_check_hip(hipMalloc(&c->scratch[slot], bytes), "hipMalloc(scratch)");
        if (st) return st;
        c->scratch_bytes[slot] = bytes;
    }
    *out = c->scratch[slot];
    return DABGPU_OK;
}
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out) { return dabgpu_scratch(c, slot, bytes, out, nullptr, false); }
int dabgpu_scratch(dabgpu_ctx* c, int slot, size_t bytes, void** out, hipStream_t user) { return dabgpu_scratch(c, slot, bytes, out, user, true); }
