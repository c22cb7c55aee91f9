// ingest.hip -- the host -> device hand-over of capture bytes.
//
// The reference's ingest is a reader thread that freads a block, converts it to complex float and calls OFDM_Demod::Process, which
// memcpy's the samples into the frame buffer (examples/app_helpers/app_ofdm_blocks.h:45-58, src/ofdm/ofdm_demodulator.cpp:550-577).
// Here the bytes cross PCIe once, in their capture format (2 B per sample for raw_u8 instead of 8), through a ring of PINNED host
// buffers with device twins: the caller fills a pinned buffer (fread straight into it), dabgpu_ingest_submit enqueues the copy on the
// pipe's own copy stream and makes the compute stream wait for it, the caller enqueues its kernels on the device twin
// (dabgpu_ofdm_demod_frames_raw, dabgpu_stream_bank_process_raw, ...) and marks it consumed.  With depth >= 2 the copy of batch
// k + 1 runs while batch k is demodulated; nothing in the steady state blocks the host except a full ring.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "dabgpu_internal.h"

struct dabgpu_ingest {
    dabgpu_ctx* ctx = nullptr;
    size_t bytes = 0;
    int depth = 0;
    int cur = -1;                                 // slot handed out by the last acquire
    hipStream_t copy_stream = nullptr;
    struct slot { void* h = nullptr; void* d = nullptr; hipEvent_t h2d = nullptr, consumed = nullptr; bool h2d_pending = false, consumed_pending = false; };
    std::vector<slot> slots;
};

extern "C" {

int dabgpu_ingest_create(dabgpu_ctx* c, size_t buffer_bytes, int depth, dabgpu_ingest** out) {
    if (!c || !out || buffer_bytes == 0 || depth < 1 || depth > 16) { dabgpu_set_error("ingest_create: invalid argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    dabgpu_ingest* g = new dabgpu_ingest();
    g->ctx = c; g->bytes = buffer_bytes; g->depth = depth;
    g->slots.resize((size_t)depth);
    int st = dabgpu_check_hip(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking), "hipStreamCreate(ingest copy stream)");
    for (int k = 0; k < depth && !st; k++) {
        auto& s = g->slots[(size_t)k];
        if ((st = dabgpu_check_hip(hipHostMalloc(&s.h, buffer_bytes, hipHostMallocDefault), "hipHostMalloc(ingest)"))) break;
        if ((st = dabgpu_check_hip(hipMalloc(&s.d, buffer_bytes), "hipMalloc(ingest)"))) break;
        if ((st = dabgpu_check_hip(hipEventCreateWithFlags(&s.h2d, hipEventDisableTiming), "hipEventCreate(ingest)"))) break;
        if ((st = dabgpu_check_hip(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming), "hipEventCreate(ingest)"))) break;
    }
    if (st) { dabgpu_ingest_destroy(g); return st; }
    *out = g;
    return DABGPU_OK;
}

void dabgpu_ingest_destroy(dabgpu_ingest* g) {
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
    for (auto& s : g->slots) {
        if (s.consumed_pending) (void)hipEventSynchronize(s.consumed);
        if (s.h2d) (void)hipEventDestroy(s.h2d);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
    }
    if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
    delete g;
}

// next pinned buffer to fill; blocks only while the copy that last read this buffer is still in flight (the ring is full)
int dabgpu_ingest_acquire(dabgpu_ingest* g, void** h_buffer) {
    if (!g || !h_buffer) { dabgpu_set_error("ingest_acquire: null argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(g->ctx);
    g->cur = (g->cur + 1) % g->depth;
    auto& s = g->slots[(size_t)g->cur];
    if (s.h2d_pending) {
        int st = dabgpu_check_hip(hipEventSynchronize(s.h2d), "hipEventSynchronize(ingest h2d)");
        if (st) return st;
        s.h2d_pending = false;
    }
    *h_buffer = s.h;
    return DABGPU_OK;
}

static dabgpu_ingest::slot* slot_of(dabgpu_ingest* g, const void* d_buffer) {
    for (auto& s : g->slots) if (s.d == d_buffer) return &s;
    return nullptr;
}

// copy the first `bytes` of the acquired buffer to its device twin: asynchronously, on the pipe's copy stream, after the kernels
// that last read the twin.  *d_buffer = the twin; no stream waits for the copy until dabgpu_ingest_wait says so
int dabgpu_ingest_submit(dabgpu_ingest* g, size_t bytes, void** d_buffer) {
    if (!g || !d_buffer || g->cur < 0 || bytes > g->bytes) { dabgpu_set_error("ingest_submit: invalid argument (acquire first; bytes <= buffer size)"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(g->ctx);
    auto& s = g->slots[(size_t)g->cur];
    int st;
    if (s.consumed_pending) {
        if ((st = dabgpu_check_hip(hipStreamWaitEvent(g->copy_stream, s.consumed, 0), "hipStreamWaitEvent(ingest consumed)"))) return st;
        s.consumed_pending = false;
    }
    if (bytes && (st = dabgpu_check_hip(hipMemcpyAsync(s.d, s.h, bytes, hipMemcpyHostToDevice, g->copy_stream), "hipMemcpyAsync(ingest)"))) return st;
    if ((st = dabgpu_check_hip(hipEventRecord(s.h2d, g->copy_stream), "hipEventRecord(ingest h2d)"))) return st;
    s.h2d_pending = true;
    *d_buffer = s.d;
    return DABGPU_OK;
}

// `compute_stream` (NULL = the default stream) waits for the copy into the twin d_buffer (call right before enqueueing the kernels that read it: a copy submitted
// early -- the next batch, while this one is still being processed -- then delays nothing that does not need it)
int dabgpu_ingest_wait(dabgpu_ingest* g, const void* d_buffer, void* compute_stream) {
    dabgpu_ingest::slot* s = g ? slot_of(g, d_buffer) : nullptr;
    if (!s) { dabgpu_set_error("ingest_wait: not a buffer of this pipe"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(g->ctx);
    return dabgpu_check_hip(hipStreamWaitEvent((hipStream_t)compute_stream, s->h2d, 0), "hipStreamWaitEvent(ingest h2d)");
}

// call after the kernels reading the twin d_buffer have been enqueued on `compute_stream`: it may be overwritten once they ran
int dabgpu_ingest_consumed(dabgpu_ingest* g, const void* d_buffer, void* compute_stream) {
    dabgpu_ingest::slot* s = g ? slot_of(g, d_buffer) : nullptr;
    if (!s) { dabgpu_set_error("ingest_consumed: not a buffer of this pipe"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(g->ctx);
    int st = dabgpu_check_hip(hipEventRecord(s->consumed, (hipStream_t)compute_stream), "hipEventRecord(ingest consumed)");
    if (st) return st;
    s->consumed_pending = true;
    return DABGPU_OK;
}

}  // extern "C"
