// ofdm_stream.hip -- device-side unsynchronised front end: a BANK of independent OFDM receivers,
// each with the state OFDM_Demod keeps between Process() calls (src/ofdm/ofdm_demodulator.h:131-176) resident in
// HBM, advanced by device kernels.  One dabgpu_stream_bank_process() = one OFDM_Demod::Process(block) for every
// stream of the bank (src/ofdm/ofdm_demodulator.cpp:235-275):
//
//   stream_l1_kernel        the L1 windows of UpdateSignalAverage (:934-950) for the whole block, one thread per window
//   stream_advance_kernel   one workgroup per stream: the level IIR, FindNullPowerDip (:291-347), ReadNullPRS (:349-358),
//                           ReadSymbols' bookkeeping (:550-562), Reset (:277-289) and the bookkeeping after a sync / a
//                           demodulated frame.  Runs until the stream needs the PRS synchroniser or the frame
//                           demodulator, or has consumed its block.  A frame that completes inside the block is NOT
//                           assembled: the demodulator reads its head from the stream's frame buffer and the rest
//                           straight from the caller's block (dabgpu_frame_desc); only the unfinished frame at the end
//                           of a block is carried over, by
//   stream_copy_kernel      many workgroups per stream, once per call after the last round
//   ofdm_sync_kernel        (ofdm_sync.hip) for the streams whose correlation window just filled
//   ofdm_demod_kernel + ofdm_phase_kernel  (ofdm_demod.hip) for the streams whose frame buffer just filled
//
// repeated ("rounds") until every stream has consumed its block; the host only reads one counter back per round
// after the rounds every stream needs in steady state.  Every float is produced by the same operations in the same
// order as the host mirror class (host/ofdm/ofdm_demodulator.cpp) and the oracle composition tests/stream_model.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"
#include "iq_decode.h"
#include "iq_sample.h"
#include "ofdm_device.h"

namespace dabgpu {

// geometry of the bank's transmission mode (src/ofdm/dab_ofdm_params_ref.cpp:11-60); mode I: 2656 / 2552 / 76 / 196608 / 5208
struct BankGeom { int mode, null_period, period, n_sym, n_fft, frame_samples, n_corr, frame_bits; };
enum { ST_FINDING_NULL = 0, ST_READING_NULL_PRS = 1, ST_RUNNING_COARSE = 2, ST_RUNNING_FINE = 3, ST_READING_SYMBOLS = 4 };
enum { PEND_NONE = 0, PEND_SYNC = 1, PEND_DEMOD = 2 };

struct StreamState {
    int state;
    float signal_avg;
    int null_start, null_end;
    int ring_index, ring_length;
    int corr_length, frame_length;
    int pending;
    int n_out;                       // frames completed in the current process() call
    int fine_time_offset;
    int total_frames_read, total_frames_desync;
    int avg_done;                    // UpdateSignalAverage already applied to the current block
    long long pos;                   // samples of the current block consumed
    // retained blocks (dabgpu_stream_bank_process_retained): frame samples [carry_dst, carry_end) of the frame under collection were
    // NOT copied into the frame buffer at the end of the previous call -- they still sit in the previous block, frame sample n at block
    // sample carry_src + (n - carry_dst).  Both bounds are even (the loaders fetch sample pairs); 0, 0 = nothing carried
    int carry_dst, carry_end;
    long long carry_src;
};

struct BankView {
    BankGeom g;
    StreamState* st;
    dabgpu_sync_state* sync;
    f2* ring;                        // [n][null_period]
    f2* corr;                        // [n][null_period + period]
    f2* frame;                       // [n][frame_samples]
    float* freq;                     // [n] coarse + fine handed to the PLL
    int* sync_active;                // [n]
    dabgpu_frame_desc* desc;         // [n] demodulation request of the round
    long long* copy_src;             // [n] carry-over copy of the call: block sample offset ...
    int* copy_dst;                   // [n] ... frame buffer offset ...
    int* copy_cnt;                   // [n] ... samples (0 = none)
    float* win;                      // [n][win_cap] L1 windows of the current block
    long long win_cap;
    int* not_done;                   // [2 lanes][8 rounds]: streams of the lane that need another round; round r counts in [r % 8] and
                                     // clears [(r + 1) % 8] (no memset between the rounds)
};


template <int SRC>
__device__ __forceinline__ float l1_window(const uint8_t* base, long long first, int k) {           // CalculateL1Average :922-932
    float acc = 0.0f;
    for (int i = 0; i < k; i++) { const f2 v = sample_at<SRC>(base, first + i); acc += __builtin_fabsf(v.x) + __builtin_fabsf(v.y); }
    return acc / (float)k;
}

template <int SRC>
__device__ __forceinline__ void copy_block_samples(f2* __restrict__ dst, const uint8_t* __restrict__ base, long long first, long long n, int t) {
    for (long long i = t; i < n; i += 256) dst[i] = sample_at<SRC>(base, first + i);
}

__device__ __forceinline__ void copy_samples(f2* __restrict__ dst, const f2* __restrict__ src, long long n, int t) {
    long long i = t;
    for (; i + 768 < n; i += 1024) {                     // four independent loads in flight
        const f2 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (; i < n; i += 256) dst[i] = src[i];
}

template <int SRC>
__global__ __launch_bounds__(256)
void stream_advance_kernel(BankView B, int s0, int n_streams, const uint8_t* __restrict__ iq, size_t stream_stride, long long n_samples,
                           dabgpu_stream_cfg cfg, int max_frames, int first_round, int ring_mode, int* __restrict__ not_done, int* __restrict__ next_not_done,
                           const uint8_t* __restrict__ prev_iq, int retain)
{
    __shared__ StreamState S;
    __shared__ float win[256];
    __shared__ int sh_i[2];
    __shared__ long long sh_w;
    const int s = s0 + blockIdx.x, t = threadIdx.x;       // streams s0 .. s0 + n_streams - 1 (a lane of the bank, see bank_process_impl)
    if (s >= s0 + n_streams) return;
    if (blockIdx.x == 0 && t == 0) *next_not_done = 0;    // the next round's counter (nobody counts in it during this round)
    if (t == 0) S = B.st[s];
    __syncthreads();
    const uint8_t* block = iq + (size_t)s * stream_stride * src_sample_bytes<SRC>::value;
    // the previous call's block, still valid (retained mode), or null: then nothing is carried
    const uint8_t* prev_block = prev_iq ? prev_iq + (size_t)s * stream_stride * src_sample_bytes<SRC>::value : nullptr;
    const int NB_NULL_PERIOD = B.g.null_period, NB_CORR = B.g.n_corr, NB_FRAME_SAMPLES = B.g.frame_samples;
    f2* ring = B.ring + (size_t)s * NB_NULL_PERIOD;
    f2* corr = B.corr + (size_t)s * NB_CORR;
    f2* frame = B.frame + (size_t)s * NB_FRAME_SAMPLES;
    const int k = cfg.signal_l1_nb_samples;

    if (first_round) {
        if (t == 0) {
            S.pos = 0; S.n_out = 0; S.avg_done = 0;
            // what the previous call left in its block instead of copying it (retained mode) is this call's carry -- unless the frame
            // cannot complete inside this block (it began in the last samples of the previous block and this block is shorter than a
            // frame: the ring form's blocks are): a frame may lie in the frame buffer, the previous block and this block, not in a
            // third block, so those few samples are copied into the frame buffer now
            const int left = B.copy_cnt[s];
            sh_i[0] = 0;
            if (prev_block != nullptr && left > 0) {
                S.carry_dst = B.copy_dst[s]; S.carry_end = B.copy_dst[s] + left; S.carry_src = B.copy_src[s];
                if ((long long)NB_FRAME_SAMPLES - (long long)S.frame_length > n_samples) sh_i[0] = 1;
            } else { S.carry_dst = 0; S.carry_end = 0; S.carry_src = 0; }
            B.copy_cnt[s] = 0;
        }
        __syncthreads();
        if (sh_i[0]) {
            copy_block_samples<SRC>(frame + S.carry_dst, prev_block, S.carry_src, (long long)(S.carry_end - S.carry_dst), t);
            __syncthreads();
            if (t == 0) { S.carry_dst = 0; S.carry_end = 0; S.carry_src = 0; }
            __syncthreads();
        }
    }

    // ---- what the previous round's sync / demod kernels produced ----
    if (S.pending == PEND_SYNC) {
        const dabgpu_sync_state y = B.sync[s];
        if (!y.sync_valid) {                                               // Reset() :277-289, :529-532
            __syncthreads();
            if (t == 0) {
                S.state = ST_FINDING_NULL; S.corr_length = 0; S.total_frames_desync++; S.fine_time_offset = 0;
                dabgpu_sync_state z = y; z.is_found_coarse = 0; z.freq_coarse = 0.0f; z.freq_fine = 0.0f; z.fine_time_offset = 0;
                B.sync[s] = z;
            }
        } else {                                                           // :536-546
            const int start = NB_NULL_PERIOD + y.fine_time_offset;
            const int count = NB_CORR - start;
            copy_samples(frame, corr + start, count, t);
            __syncthreads();
            if (t == 0) { S.frame_length = count; S.corr_length = 0; S.fine_time_offset = y.fine_time_offset; S.state = ST_READING_SYMBOLS; }
        }
        if (t == 0) { S.pending = PEND_NONE; B.sync_active[s] = 0; }
        __syncthreads();
    }
    // (a demodulation request never stays pending: the frame's bookkeeping is done where the request is made, see ReadSymbols below)
    if (t == 0) B.desc[s].slot = -1;
    __syncthreads();

    // ---- UpdateSignalAverage over the whole block, once per process() (:934-950); windows from stream_l1_kernel ----
    if (!S.avg_done) {
        if (n_samples >= k) {
            const long long stride = (long long)k * cfg.signal_l1_nb_decimate;
            const long long n_win = (n_samples - k + stride - 1) / stride;         // i = 0, stride, ... < n - k
            const float* wsrc = B.win + (size_t)s * B.win_cap;
            for (long long w0 = 0; w0 < n_win; w0 += 256) {
                if (w0 + t < n_win) win[t] = wsrc[w0 + t];
                __syncthreads();
                if (t == 0) {
                    const int m = (int)((n_win - w0 < 256) ? (n_win - w0) : 256);
                    float a = S.signal_avg;
                    const float beta = cfg.signal_l1_update_beta;
                    for (int j = 0; j < m; j++) a = beta * a + (1.0f - beta) * win[j];
                    S.signal_avg = a;
                }
                __syncthreads();
            }
        }
        if (t == 0) S.avg_done = 1;
        __syncthreads();
    }
    // ---- the Process() loop (:241-274) up to the next device-kernel request ----
    while (S.pos < n_samples && S.pending == PEND_NONE) {
        const long long bpos = S.pos;                                               // buf[i] of the reference = block sample bpos + i
        const long long rest = n_samples - S.pos;
        if (S.state == ST_FINDING_NULL) {
            // FindNullPowerDip :291-347
            const float start_thresh = S.signal_avg * cfg.thresh_null_start;
            const float end_thresh = S.signal_avg * cfg.thresh_null_end;
            const long long n_win = (rest > k) ? (rest - 1) / k : 0;                // i = 0, k, ... while i < rest - k
            if (t == 0) { sh_i[0] = S.null_start; sh_i[1] = 0; }                    // [1] = null end found
            long long nb_read = rest;
            __syncthreads();
            for (long long w0 = 0; w0 < n_win; w0 += 256) {
                const long long w = w0 + t;
                if (w < n_win) win[t] = l1_window<SRC>(block, bpos + w * k, k);
                __syncthreads();
                if (t == 0) {
                    const int m = (int)((n_win - w0 < 256) ? (n_win - w0) : 256);
                    int started = sh_i[0];
                    for (int j = 0; j < m; j++) {
                        const float l1 = win[j];
                        if (started) {
                            if (l1 > end_thresh) { sh_i[1] = 1; sh_w = w0 + j; break; }
                        } else if (l1 < start_thresh) {
                            started = 1;
                        }
                    }
                    sh_i[0] = started;
                }
                __syncthreads();
                if (sh_i[1]) break;
            }
            const int null_end = sh_i[1];
            if (null_end) nb_read = sh_w * k + k;
            // circular buffer of the last nb_null_period samples (:325, circular_buffer.h:18-37)
            const int cap = NB_NULL_PERIOD;
            const long long keep = (nb_read < cap) ? nb_read : cap;
            const long long first = nb_read - keep;
            const int ring_index = S.ring_index;
            for (long long j = t; j < keep; j += 256) ring[(int)((ring_index + first + j) % cap)] = sample_at<SRC>(block, bpos + first + j);
            __syncthreads();
            const int new_index = (int)((ring_index + nb_read) % cap);
            const int new_length = (int)(((long long)S.ring_length + nb_read < cap) ? (S.ring_length + nb_read) : cap);
            if (null_end) {                                                         // :333-338
                for (int j = t; j < new_length; j += 256) corr[j] = ring[(j + new_index) % cap];
            }
            __syncthreads();
            if (t == 0) {
                S.ring_index = new_index; S.ring_length = new_length; S.null_start = sh_i[0];
                if (null_end) { S.corr_length = new_length; S.null_start = 0; S.null_end = 0; S.ring_length = 0; S.state = ST_READING_NULL_PRS; }
                S.pos += nb_read;
            }
            __syncthreads();
        } else if (S.state == ST_READING_NULL_PRS) {
            // ReadNullPRS :349-358
            const long long want = NB_CORR - S.corr_length;
            const long long take = (want < rest) ? want : rest;
            copy_block_samples<SRC>(corr + S.corr_length, block, bpos, take, t);
            __syncthreads();
            if (t == 0) {
                S.corr_length += (int)take; S.pos += take;
                // like the reference, a window that fills with the last sample of a block is synchronised by the NEXT Process()
                if (S.corr_length == NB_CORR) {
                    S.state = ST_RUNNING_COARSE;
                    if (S.pos < n_samples) { S.pending = PEND_SYNC; B.sync_active[s] = 1; }
                }
            }
            __syncthreads();
        } else if (S.state == ST_READING_SYMBOLS) {
            // ReadSymbols :550-577
            const int have = S.frame_length;
            const long long want = NB_FRAME_SAMPLES - have;
            const bool full = (want <= rest);
            const long long take = full ? want : rest;
            const int cd = S.carry_dst, ce = S.carry_end;                       // (retained mode) part of [0, have) still in the previous block
            const long long csrc = S.carry_src;
            if (full) {
                // the frame completes inside this block: no assembly, the demodulator reads [0, split) from the frame
                // buffer (or, retained mode, the previous block) and the rest from the block; one sample is moved over when the boundary
                // would split a pair
                if ((have & 1) && t == 0) frame[have] = sample_at<SRC>(block, bpos);
                // the NULL symbol at the end of this frame heads the next correlation window (:558-562)
                const int null_at = B.g.n_sym * B.g.period;
                for (int j = t; j < NB_NULL_PERIOD; j += 256) {
                    const int idx = null_at + j;
                    corr[j] = (idx < have) ? ((idx >= cd && idx < ce) ? sample_at<SRC>(prev_block, csrc + (idx - cd)) : frame[idx])
                                           : sample_at<SRC>(block, bpos + (idx - have));
                }
            } else if (t == 0) {
                // the block ends inside this frame: its samples so far are carried over into the stream's frame buffer by
                // stream_copy_kernel, ONCE per call after the last round -- a stream that asks for it has consumed its block and takes
                // no further part in the rounds (a copy launch per round, empty in all rounds but the last, cost 67 us each).
                // Retained mode: no copy at all -- the record becomes the next call's carry; its bounds are made even here (the
                // demodulator loads sample pairs), the odd samples at either end go to the frame buffer now.
                int a = have, e = have + (int)take;
                long long src = S.pos;
                if (retain) {
                    if ((a & 1) && a < e) { frame[a] = sample_at<SRC>(block, src); a++; src++; }
                    if ((e & 1) && a < e) { frame[e - 1] = sample_at<SRC>(block, S.pos + take - 1); e--; }
                }
                B.copy_src[s] = src; B.copy_dst[s] = a; B.copy_cnt[s] = e - a;
            }
            __syncthreads();
            if (t == 0) {
                const long long pos0 = S.pos;
                S.frame_length += (int)take; S.pos += take;
                if (full) {
                    S.corr_length = NB_NULL_PERIOD;
                    const dabgpu_sync_state y = B.sync[s];
                    B.freq[s] = y.freq_coarse + y.freq_fine;
                    // a frame beyond the caller's capacity is demodulated into the last slot (and reported through n_frames)
                    // (ring mode: slot of the stream's frame-history ring = frames demodulated so far mod ring length)
                    const int slot = ring_mode ? (S.total_frames_read % max_frames) : ((S.n_out < max_frames) ? S.n_out : (max_frames - 1));
                    dabgpu_frame_desc d;
                    d.slot = s * max_frames + slot;
                    d.split = have + (have & 1);
                    d.tail_off = pos0 + (have & 1);
                    d.carry_dst = cd; d.carry_end = ce; d.carry_off = csrc - cd;
                    B.desc[s] = d;
                    S.carry_dst = 0; S.carry_end = 0; S.carry_src = 0;
                    // The demodulation of this frame runs in THIS round (the round's kernel order is advance, copy, demod, phase, sync),
                    // so the state machine does the frame's bookkeeping now (:563-576) and reads on into the next NULL + PRS window: one
                    // round per frame instead of two.  The reference's order is kept -- the fine-frequency update of frame k - 1 (phase
                    // kernel) precedes the synchronisation of frame k (sync kernel) inside the round.
                    S.total_frames_read++; S.n_out++; S.frame_length = 0; S.state = ST_READING_NULL_PRS;
                }
            }
            __syncthreads();
        } else {
            // RUNNING_*: the correlation window filled at the very end of the previous block (:247-262)
            __syncthreads();
            if (t == 0) { S.pending = PEND_SYNC; B.sync_active[s] = 1; }
            __syncthreads();
        }
    }
    if (t == 0) {
        B.st[s] = S;
        if (S.pos < n_samples || S.pending != PEND_NONE) atomicAdd(not_done, 1);
    }
}

// L1 windows of UpdateSignalAverage for every stream's block: window w covers samples [w*stride, w*stride + k).
// 64 windows per workgroup: the samples are fetched coalesced (consecutive lanes = consecutive samples of a window) and
// reduced to |re| + |im| into LDS; one lane per window then adds its k terms in the reference's sequential order.
constexpr int L1_WINDOWS = 64;
constexpr int L1_MAX_K = 128;
template <int SRC>
__global__ __launch_bounds__(256)
void stream_l1_kernel(BankView B, const uint8_t* __restrict__ iq, size_t stream_stride, long long n_win, int k, long long stride) {
    __shared__ float term[L1_WINDOWS * (L1_MAX_K + 1)];
    const long long w0 = (long long)blockIdx.x * L1_WINDOWS;
    const int s = blockIdx.y, t = threadIdx.x;
    const uint8_t* base = iq + (size_t)s * stream_stride * src_sample_bytes<SRC>::value;
    const int m = (int)((n_win - w0 < L1_WINDOWS) ? (n_win - w0) : L1_WINDOWS);
    if (k <= L1_MAX_K) {
        for (int e = t; e < m * k; e += 256) {
            const int w = e / k, i = e - w * k;
            const f2 v = sample_at<SRC>(base, (w0 + w) * stride + i);
            term[w * (k + 1) + i] = __builtin_fabsf(v.x) + __builtin_fabsf(v.y);
        }
        __syncthreads();
        if (t < m) {
            float acc = 0.0f;
            for (int i = 0; i < k; i++) acc += term[t * (k + 1) + i];
            B.win[(size_t)s * B.win_cap + w0 + t] = acc / (float)k;
        }
    } else if (t < m) {
        B.win[(size_t)s * B.win_cap + w0 + t] = l1_window<SRC>(base, (w0 + t) * stride, k);
    }
}

// the carry-over copy block -> frame buffer (the unfinished frame at the end of a block), COPY_WGS workgroups per stream, once per call
constexpr int COPY_WGS = 16;
template <int SRC>
__global__ __launch_bounds__(256)
void stream_copy_kernel(BankView B, int s0, const uint8_t* __restrict__ iq, size_t stream_stride) {
    const int s = s0 + blockIdx.y;
    const int cnt = B.copy_cnt[s];
    if (cnt == 0) return;
    const uint8_t* src = iq + (size_t)s * stream_stride * src_sample_bytes<SRC>::value;
    const long long first = B.copy_src[s];
    f2* dst = B.frame + (size_t)s * B.g.frame_samples + B.copy_dst[s];
    const int per = (((cnt + COPY_WGS - 1) / COPY_WGS) + 255) & ~255;
    const int a = blockIdx.x * per;
    const int b = (a + per < cnt) ? (a + per) : cnt;
    int i = a + threadIdx.x;
    for (; i + 3 * 256 < b; i += 4 * 256) {                   // four independent loads in flight
        const f2 v0 = sample_at<SRC>(src, first + i), v1 = sample_at<SRC>(src, first + i + 256), v2 = sample_at<SRC>(src, first + i + 512),
                 v3 = sample_at<SRC>(src, first + i + 768);
        dst[i] = v0; dst[i + 256] = v1; dst[i + 512] = v2; dst[i + 768] = v3;
    }
    for (; i < b; i += 256) dst[i] = sample_at<SRC>(src, first + i);
}

__global__ void stream_report_kernel(BankView B, int n_streams, int* __restrict__ n_frames, dabgpu_stream_status* __restrict__ status,
                                     int ring_frames) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const StreamState S = B.st[s];
    if (n_frames) n_frames[s] = (ring_frames > 0) ? ((S.n_out > 0) ? ((S.total_frames_read - 1) % ring_frames) : -1) : S.n_out;
    if (status) {
        const dabgpu_sync_state y = B.sync[s];
        dabgpu_stream_status o;
        o.state = S.state; o.signal_l1_average = S.signal_avg; o.freq_coarse = y.freq_coarse; o.freq_fine = y.freq_fine;
        o.is_found_coarse = y.is_found_coarse; o.fine_time_offset = S.fine_time_offset;
        o.total_frames_read = S.total_frames_read; o.total_frames_desync = S.total_frames_desync;
        status[s] = o;
    }
}

}  // namespace dabgpu

using namespace dabgpu;

struct dabgpu_stream_bank {
    dabgpu_ctx* ctx = nullptr;
    size_t n = 0;
    dabgpu_stream_cfg cfg{};
    BankView view{};
    float* d_corr_out = nullptr;       // [n][n_sym][2] cyclic-prefix correlations of the last demodulated frames
    dabgpu_stream_status* d_status = nullptr;
    float* d_raw_scratch = nullptr;    // converted block of dabgpu_stream_bank_process_raw (grow-only)
    size_t raw_scratch_bytes = 0;
    std::vector<void*> allocs;
    hipStream_t side = nullptr;        // second lane of a call (mode I banks of >= 1024 streams): half of the streams run their rounds here
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // retained blocks: the last call left the unfinished frames' samples in ITS block (no carry-over copy); the next call must be handed
    // that block again (dabgpu_stream_bank_process_retained), or dabgpu_stream_bank_release must copy them out of it first
    bool carry_pending = false;
    int carry_src = 0;                 // loader kind (SRC) and stream stride of that block
    size_t carry_stride = 0;
};

extern "C" {

void dabgpu_stream_bank_destroy(dabgpu_stream_bank* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : b->allocs) (void)hipFree(p);
    if (b->side) (void)hipStreamDestroy(b->side);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    delete b;
}

int dabgpu_stream_bank_reset(dabgpu_stream_bank* b, void* stream) {
    if (!b) { dabgpu_set_error("stream_bank_reset: null bank"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(b->ctx);
    hipStream_t s = (hipStream_t)stream;
    int st;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    CK(hipMemsetAsync(b->view.st, 0, b->n * sizeof(StreamState), s));
    CK(hipMemsetAsync(b->view.sync, 0, b->n * sizeof(dabgpu_sync_state), s));
    CK(hipMemsetAsync(b->view.sync_active, 0, b->n * sizeof(int), s));
    CK(hipMemsetAsync(b->view.desc, 0xFF, b->n * sizeof(dabgpu_frame_desc), s));
    CK(hipMemsetAsync(b->view.ring, 0, b->n * (size_t)b->view.g.null_period * sizeof(f2), s));
    CK(hipMemsetAsync(b->view.corr, 0, b->n * (size_t)b->view.g.n_corr * sizeof(f2), s));
    CK(hipMemsetAsync(b->view.copy_cnt, 0, b->n * sizeof(int), s));
    CK(hipMemsetAsync(b->view.not_done, 0, 16 * sizeof(int), s));
#undef CK
    b->carry_pending = false;
    return DABGPU_OK;
}

int dabgpu_stream_bank_create(dabgpu_ctx* c, size_t n_streams, const dabgpu_stream_cfg* cfg, dabgpu_stream_bank** out) {
    return dabgpu_stream_bank_create_mode(c, 1, n_streams, cfg, out);
}

int dabgpu_stream_bank_create_mode(dabgpu_ctx* c, int mode, size_t n_streams, const dabgpu_stream_cfg* cfg, dabgpu_stream_bank** out) {
    if (!c || !out || n_streams == 0 || n_streams > (size_t)(1 << 20)) { dabgpu_set_error("stream_bank_create: invalid argument"); return DABGPU_ERR_INVALID_ARG; }
    ModeGeom mg;
    if (!mode_geometry(mode, mg)) { dabgpu_set_error("stream_bank_create: invalid transmission mode %d", mode); return DABGPU_ERR_INVALID_ARG; }
    dabgpu_stream_cfg k;
    if (cfg) k = *cfg; else dabgpu_stream_cfg_default(&k);
    if (k.signal_l1_nb_samples <= 0 || k.signal_l1_nb_decimate <= 0) { dabgpu_set_error("stream_bank_create: invalid L1 window"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    dabgpu_stream_bank* b = new (std::nothrow) dabgpu_stream_bank();
    if (!b) return DABGPU_ERR_HIP;
    b->ctx = c; b->n = n_streams; b->cfg = k;
    BankGeom& G = b->view.g;
    G.mode = mode; G.null_period = mg.null_period; G.period = mg.period; G.n_sym = mg.n_sym; G.n_fft = mg.n_fft;
    G.frame_samples = mg.frame_samples; G.n_corr = mg.null_period + mg.period; G.frame_bits = mg.frame_bits;
    int st = DABGPU_OK;
    auto alloc = [&](void** p, size_t bytes) {
        if (st) return;
        st = dabgpu_check_hip(hipMalloc(p, bytes), "hipMalloc(stream bank)");
        if (!st) b->allocs.push_back(*p);
    };
    alloc((void**)&b->view.st, n_streams * sizeof(StreamState));
    alloc((void**)&b->view.sync, n_streams * sizeof(dabgpu_sync_state));
    alloc((void**)&b->view.ring, n_streams * (size_t)G.null_period * sizeof(f2));
    alloc((void**)&b->view.corr, n_streams * (size_t)G.n_corr * sizeof(f2));
    alloc((void**)&b->view.frame, n_streams * (size_t)G.frame_samples * sizeof(f2));
    alloc((void**)&b->view.freq, n_streams * sizeof(float));
    alloc((void**)&b->view.sync_active, n_streams * sizeof(int));
    alloc((void**)&b->view.desc, n_streams * sizeof(dabgpu_frame_desc));
    alloc((void**)&b->view.copy_src, n_streams * sizeof(long long));
    alloc((void**)&b->view.copy_dst, n_streams * sizeof(int));
    alloc((void**)&b->view.copy_cnt, n_streams * sizeof(int));
    alloc((void**)&b->view.not_done, 16 * sizeof(int));
    alloc((void**)&b->d_corr_out, n_streams * (size_t)G.n_sym * 2 * sizeof(float));
    alloc((void**)&b->d_status, n_streams * sizeof(dabgpu_stream_status));
    if (!st) st = dabgpu_check_hip(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking), "hipStreamCreate(stream bank)");
    if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming), "hipEventCreate");
    if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming), "hipEventCreate");
    if (!st) st = dabgpu_stream_bank_reset(b, c->stream);
    if (!st) st = dabgpu_check_hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st) { dabgpu_stream_bank_destroy(b); return st; }
    *out = b;
    return DABGPU_OK;
}

}  // extern "C"

// ring_mode: d_bits = [n_streams][max_frames_per_stream = ring length][230400], a completed frame goes to slot
// (frames demodulated so far) mod ring length, d_n_frames receives the slot written in this call or -1
template <int SRC>
static int bank_process_impl(dabgpu_stream_bank* b, const void* d_iq, size_t stream_stride_samples, size_t n_samples,
                             int8_t* d_bits, size_t max_frames_per_stream, int32_t* d_n_frames, void* stream, int ring_mode = 0,
                             int classed = 0, const void* d_prev = nullptr, int retain = 0) {
    if (!b || !d_iq || !d_bits) { dabgpu_set_error("stream_bank_process: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_samples == 0) return DABGPU_OK;
    if (b->carry_pending && (!d_prev || b->carry_src != SRC || b->carry_stride != stream_stride_samples)) {
        dabgpu_set_error("stream_bank_process: the previous call retained its block -- pass that block (same format and stride) to "
                         "dabgpu_stream_bank_process_retained, or call dabgpu_stream_bank_release first");
        return DABGPU_ERR_INVALID_ARG;
    }
    const BankGeom& G = b->view.g;
    const int NB_FRAME_SAMPLES = G.frame_samples, NB_CORR = G.n_corr, NB_NULL_PERIOD = G.null_period;
    // a frame needs at least n_sym symbols + a NULL minus the sync's pull-back of new samples
    const size_t need = ring_mode ? 1 : (n_samples / (size_t)(NB_FRAME_SAMPLES - NB_CORR) + 2);
    if (ring_mode && n_samples > (size_t)(NB_FRAME_SAMPLES - NB_CORR)) {
        dabgpu_set_error("stream_bank_process_ring: at most %d samples per call (one frame per stream per call)", NB_FRAME_SAMPLES - NB_CORR);
        return DABGPU_ERR_INVALID_ARG;
    }
    if (max_frames_per_stream < need || max_frames_per_stream > (size_t)(1 << 20) || b->n * max_frames_per_stream > (size_t)0x7FFFFFFF) {
        dabgpu_set_error("stream_bank_process: max_frames_per_stream must be at least %zu for %zu samples", need, n_samples);
        return DABGPU_ERR_INVALID_ARG;
    }
    if (((uintptr_t)d_iq & (src_sample_bytes<SRC>::value - 1)) || ((uintptr_t)d_bits & 15)) { dabgpu_set_error("stream_bank_process: misaligned buffer"); return DABGPU_ERR_INVALID_ARG; }
    dabgpu_ctx* c = b->ctx;
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)b->n;
    int st;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    // Retained blocks (mode I): the unfinished frame at the end of a block is not copied into the stream's frame buffer -- the next
    // call reads it where it is.  A frame may then lie in the frame buffer, the previous block and the current block.  With blocks of
    // a frame's length or more every pending frame completes inside the next block; with the ring form's blocks (at most a frame
    // minus a correlation window) the few streams whose frame began in the last samples of the previous block copy those samples
    // themselves at the start of the call (stream_advance_kernel); a shorter block copies all carried samples first, like a release.
    const bool use_retain = retain && G.mode == 1;
    const bool prev_valid = b->carry_pending && n_samples >= (size_t)(NB_FRAME_SAMPLES - NB_CORR);
    if (b->carry_pending && !prev_valid) {
        hipLaunchKernelGGL(stream_copy_kernel<SRC>, dim3(COPY_WGS, (unsigned)n), dim3(256), 0, s, b->view, 0, static_cast<const uint8_t*>(d_prev),
                           stream_stride_samples);
        CK(hipGetLastError());
    }
    b->carry_pending = false;
    const uint8_t* prev_iq = prev_valid ? static_cast<const uint8_t*>(d_prev) : nullptr;
    // L1 windows of the signal-level IIR for every stream (grow-only scratch)
    const int k = b->cfg.signal_l1_nb_samples;
    if (n_samples >= (size_t)k) {
        const long long stride = (long long)k * b->cfg.signal_l1_nb_decimate;
        const long long n_win = ((long long)n_samples - k + stride - 1) / stride;
        if (n_win > b->view.win_cap) {
            CK(hipStreamSynchronize(s));
            if (b->view.win) { (void)hipFree(b->view.win); b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)b->view.win)); }
            b->view.win = nullptr; b->view.win_cap = 0;
            CK(hipMalloc((void**)&b->view.win, (size_t)n * (size_t)n_win * sizeof(float)));
            b->allocs.push_back(b->view.win);
            b->view.win_cap = n_win;
        }
        if (n_win > 0) {
            hipLaunchKernelGGL(stream_l1_kernel<SRC>, dim3((unsigned)((n_win + L1_WINDOWS - 1) / L1_WINDOWS), (unsigned)n), dim3(256), 0, s,
                               b->view, static_cast<const uint8_t*>(d_iq), stream_stride_samples, n_win, k, stride);
            CK(hipGetLastError());
        }
    }
    const float *d_prs = nullptr, *d_prs_time_ref = nullptr;               // PRS spectrum / coarse-sync reference of the bank's mode
    if ((st = dabgpu_mode_sync_tables(c, G.mode, &d_prs, &d_prs_time_ref))) return st;
    // rounds every locked stream needs: one per frame that can complete inside this block (the demodulation of frame k - 1 and the
    // synchronisation of frame k share a round) plus one for the partial frame at the end; a stream that needs more (re-acquisition)
    // is caught by the counter read back after these
    const int blind_rounds = (int)((n_samples + NB_FRAME_SAMPLES - 1) / NB_FRAME_SAMPLES) + 1;
    // Two lanes: the rounds of a stream are a serial chain (advance -> copy -> demodulate frame k - 1 -> phase -> synchronise frame k
    // -> advance ...) of one bandwidth-sized kernel and four latency-bound small ones, but streams are independent -- a mode I bank
    // of >= 1024 streams runs the two halves of its streams on two HIP streams, so that one half's small kernels run beside the other
    // half's demodulation.  Same kernels, same per-stream order, same results.
    const int n_lanes = (G.mode == 1 && n >= 1024) ? 2 : 1;      // (each half must still fill the chip: 256 streams lost 10-30 % in two lanes)
    hipStream_t lane_stream[2] = {s, b->side};
    const int lane_lo[3] = {0, (n_lanes == 2) ? n / 2 : n, n};
    if (n_lanes == 2) {
        CK(hipEventRecord(b->ev_fork, s));
        CK(hipStreamWaitEvent(b->side, b->ev_fork, 0));
    }
    // (an error inside the rounds must not leave work on the side stream unordered with the caller's stream: the caller may free the
    // block and the output buffers as soon as the call has returned)
#define CKL(call) do { st = dabgpu_check_hip((call), #call); if (st) { if (n_lanes == 2) (void)hipStreamSynchronize(b->side); return st; } } while (0)
    const size_t sb = src_sample_bytes<SRC>::value;
    // run length of the bank's demodulation: four workgroups per frame (19 + 19 + 19 + 18 symbols).  Shorter-lived workgroups hand
    // their slots to the round's small kernels sooner: 256 streams +8 % against three per frame, 1024 streams +1 %; six per frame loses again
    const int bank_spb = 19;
    int h_not_done[2] = {1, 0};
    for (int round = 0; h_not_done[0] + h_not_done[1] != 0; round++) {
        for (int l = 0; l < n_lanes; l++) {
            hipStream_t ls = lane_stream[l];
            const int s0 = lane_lo[l], cnt = lane_lo[l + 1] - lane_lo[l];
            const uint8_t* iq_l = static_cast<const uint8_t*>(d_iq) + (size_t)s0 * stream_stride_samples * sb;
            if (round == 0) CKL(hipMemsetAsync(b->view.not_done + 8 * l, 0, sizeof(int), ls));
            hipLaunchKernelGGL(stream_advance_kernel<SRC>, dim3((unsigned)cnt), dim3(256), 0, ls, b->view, s0, cnt, static_cast<const uint8_t*>(d_iq),
                               stream_stride_samples, (long long)n_samples, b->cfg, (int)max_frames_per_stream, round == 0 ? 1 : 0, ring_mode,
                               b->view.not_done + 8 * l + (round & 7), b->view.not_done + 8 * l + ((round + 1) & 7), prev_iq, use_retain ? 1 : 0);
            CKL(hipGetLastError());
            float* corr_l = b->d_corr_out + (size_t)s0 * G.n_sym * 2;
            if (G.mode == 1) {
                CKL(dabgpu_launch_ofdm_demod(b->view.frame + (size_t)s0 * G.frame_samples, SRC, b->view.freq + s0, d_bits, corr_l, nullptr, nullptr,
                                            c->d_tw, c->d_inv_map, cnt, bank_spb, 0, b->view.desc + s0, iq_l, stream_stride_samples, classed, ls, nullptr, nullptr, 0.0f,
                                            prev_iq ? prev_iq + (size_t)s0 * stream_stride_samples * sb : nullptr));
            } else if ((st = dabgpu_launch_ofdm_demod_mode(     /* (modes II-IV run in one lane: n_lanes == 1, s0 == 0, cnt == n) */c, G.mode, b->view.frame, SRC, b->view.freq, d_bits, b->d_corr_out, nullptr, n, 0,
                                                           b->view.desc, d_iq, stream_stride_samples, ls))) {
                if (n_lanes == 2) (void)hipStreamSynchronize(b->side);
                return st;
            }
            CKL(dabgpu_launch_ofdm_phase(corr_l, cnt, b->cfg.sync.fine_freq_update_beta, nullptr, &b->view.sync[s0].freq_fine,
                                        (int)(sizeof(dabgpu_sync_state) / sizeof(float)), b->view.desc + s0, G.n_sym, G.n_fft, ls));
            // (after the phase kernel: the synchroniser of frame k sees the fine frequency the phase of frame k - 1 left, as in the reference)
            CKL(dabgpu_launch_sync(reinterpret_cast<const float*>(b->view.corr + (size_t)s0 * NB_CORR + NB_NULL_PERIOD), NB_CORR, cnt, &b->cfg.sync,
                                  b->view.sync + s0, nullptr, nullptr, c->d_tw, d_prs, d_prs_time_ref, b->view.sync_active + s0, G.mode, ls));
        }
        if (round + 1 >= blind_rounds) {
            for (int l = 0; l < n_lanes; l++) CKL(hipMemcpyAsync(&h_not_done[l], b->view.not_done + 8 * l + (round & 7), sizeof(int), hipMemcpyDeviceToHost, lane_stream[l]));
            for (int l = 0; l < n_lanes; l++) CKL(hipStreamSynchronize(lane_stream[l]));
        }
        if (round > (1 << 22)) { dabgpu_set_error("stream_bank_process: no progress"); if (n_lanes == 2) (void)hipStreamSynchronize(b->side); return DABGPU_ERR_HIP; }
    }
    for (int l = 0; l < n_lanes && !use_retain; l++) {                    // the carry-over copies of the call, every lane its own streams
        hipLaunchKernelGGL(stream_copy_kernel<SRC>, dim3(COPY_WGS, (unsigned)(lane_lo[l + 1] - lane_lo[l])), dim3(256), 0, lane_stream[l], b->view,
                           lane_lo[l], static_cast<const uint8_t*>(d_iq), stream_stride_samples);
        CKL(hipGetLastError());
    }
#undef CKL
    if (n_lanes == 2) {
        CK(hipEventRecord(b->ev_join, b->side));
        CK(hipStreamWaitEvent(s, b->ev_join, 0));
    }
    if (d_n_frames) {
        hipLaunchKernelGGL(stream_report_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, b->view, n, d_n_frames,
                           (dabgpu_stream_status*)nullptr, ring_mode ? (int)max_frames_per_stream : 0);
        CK(hipGetLastError());
    }
#undef CK
    if (use_retain) { b->carry_pending = true; b->carry_src = SRC; b->carry_stride = stream_stride_samples; }
    return DABGPU_OK;
}

// dabgpu_stream_bank_release: copies what the last retained call left in its block into the frame buffers
template <int SRC>
static int bank_release_impl(dabgpu_stream_bank* b, const void* d_prev, size_t stream_stride_samples, void* stream) {
    if (b->carry_src != SRC || b->carry_stride != stream_stride_samples) {
        dabgpu_set_error("stream_bank_release: not the format / stride of the retained block"); return DABGPU_ERR_INVALID_ARG;
    }
    DABGPU_BIND(b->ctx);
    hipLaunchKernelGGL(stream_copy_kernel<SRC>, dim3(COPY_WGS, (unsigned)b->n), dim3(256), 0, (hipStream_t)stream, b->view, 0,
                       static_cast<const uint8_t*>(d_prev), stream_stride_samples);
    const int st = dabgpu_check_hip(hipGetLastError(), "stream_copy_kernel launch");
    if (st) return st;
    // (the copy records stay set; the next call's first round clears them -- it is handed no previous block, so nothing is carried)
    b->carry_pending = false;
    return DABGPU_OK;
}

extern "C" {

int dabgpu_stream_bank_process_retained(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                        const void* d_prev_raw, int8_t* d_bits, size_t max_frames_per_stream, int32_t* d_n_frames, void* stream) {
    if (!b || !d_raw) { dabgpu_set_error("stream_bank_process_retained: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_samples == 0) return DABGPU_OK;
    switch (format) {
    case DABGPU_IQ_RAW_F32L: case DABGPU_IQ_WAV_F32:
        return bank_process_impl<0>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream, 0, 0, d_prev_raw, 1);
    case DABGPU_IQ_RAW_U8: case DABGPU_IQ_WAV_PCM8:
        return bank_process_impl<1>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream, 0, 0, d_prev_raw, 1);
    case DABGPU_IQ_RAW_S8:
        return bank_process_impl<2>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream, 0, 0, d_prev_raw, 1);
    case DABGPU_IQ_RAW_S16L: case DABGPU_IQ_WAV_PCM16:
        return bank_process_impl<3>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream, 0, 0, d_prev_raw, 1);
    default: break;
    }
    dabgpu_set_error("stream_bank_process_retained: format %d is not read by the bank's kernels directly (raw_f32l, raw_u8, raw_s8, raw_s16l, wav PCM8 / PCM16 / float32)", format);
    return DABGPU_ERR_INVALID_ARG;
}

int dabgpu_stream_bank_release(dabgpu_stream_bank* b, const void* d_prev_raw, int format, size_t stream_stride_samples, void* stream) {
    if (!b) { dabgpu_set_error("stream_bank_release: null bank"); return DABGPU_ERR_INVALID_ARG; }
    if (!b->carry_pending) return DABGPU_OK;
    if (!d_prev_raw) { dabgpu_set_error("stream_bank_release: null block"); return DABGPU_ERR_INVALID_ARG; }
    switch (format) {
    case DABGPU_IQ_RAW_F32L: case DABGPU_IQ_WAV_F32: return bank_release_impl<0>(b, d_prev_raw, stream_stride_samples, stream);
    case DABGPU_IQ_RAW_U8: case DABGPU_IQ_WAV_PCM8: return bank_release_impl<1>(b, d_prev_raw, stream_stride_samples, stream);
    case DABGPU_IQ_RAW_S8: return bank_release_impl<2>(b, d_prev_raw, stream_stride_samples, stream);
    case DABGPU_IQ_RAW_S16L: case DABGPU_IQ_WAV_PCM16: return bank_release_impl<3>(b, d_prev_raw, stream_stride_samples, stream);
    default: break;
    }
    dabgpu_set_error("stream_bank_release: not the format of the retained block");
    return DABGPU_ERR_INVALID_ARG;
}

int dabgpu_stream_bank_process(dabgpu_stream_bank* b, const float* d_iq, size_t stream_stride_samples, size_t n_samples,
                               int8_t* d_bits, size_t max_frames_per_stream, int32_t* d_n_frames, void* stream) {
    return bank_process_impl<0>(b, d_iq, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
}

// the same from blocks still in their capture format: dequantised on the device (iq_convert_kernel, reader arithmetic of
// examples/app_helpers/app_iq_readers.h) into bank-owned scratch on `stream`, then processed as above
int dabgpu_stream_bank_process_raw(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                   int8_t* d_bits, size_t max_frames_per_stream, int32_t* d_n_frames, void* stream) {
    if (!b || !d_raw) { dabgpu_set_error("stream_bank_process_raw: null argument"); return DABGPU_ERR_INVALID_ARG; }
    const size_t sb = dabgpu_iq_format_sample_bytes(format);
    if (sb == 0) { dabgpu_set_error("stream_bank_process_raw: unknown format %d", format); return DABGPU_ERR_INVALID_ARG; }
    if (n_samples == 0) return DABGPU_OK;
    // formats the kernels read by themselves: no conversion pass, 2-4 bytes per sample from HBM instead of 8
    switch (format) {
    case DABGPU_IQ_RAW_F32L: case DABGPU_IQ_WAV_F32:
        return bank_process_impl<0>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
    case DABGPU_IQ_RAW_U8: case DABGPU_IQ_WAV_PCM8:
        return bank_process_impl<1>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
    case DABGPU_IQ_RAW_S8:
        return bank_process_impl<2>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
    case DABGPU_IQ_RAW_S16L: case DABGPU_IQ_WAV_PCM16:
        return bank_process_impl<3>(b, d_raw, stream_stride_samples, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
    default: break;
    }
    if (((stream_stride_samples * sb) & 15) != 0 && b->n > 1) {
        dabgpu_set_error("stream_bank_process_raw: the byte stride between streams must be a multiple of 16"); return DABGPU_ERR_INVALID_ARG;
    }
    DABGPU_BIND(b->ctx);
    hipStream_t s = (hipStream_t)stream;
    const size_t padded = (n_samples + 1) & ~(size_t)1;                    // keeps every stream's converted block 16-byte aligned
    const size_t need = b->n * padded * 2 * sizeof(float);
    if (need > b->raw_scratch_bytes) {
        int st = dabgpu_check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        if (st) return st;
        if (b->d_raw_scratch) { (void)hipFree(b->d_raw_scratch); b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)b->d_raw_scratch)); }
        b->d_raw_scratch = nullptr; b->raw_scratch_bytes = 0;
        if ((st = dabgpu_check_hip(hipMalloc((void**)&b->d_raw_scratch, need), "hipMalloc(stream bank raw scratch)"))) return st;
        b->allocs.push_back(b->d_raw_scratch);
        b->raw_scratch_bytes = need;
    }
    for (size_t k = 0; k < b->n; k++) {
        const int st = dabgpu_iq_convert(b->ctx, static_cast<const uint8_t*>(d_raw) + k * stream_stride_samples * sb, format, n_samples,
                                         b->d_raw_scratch + k * padded * 2, stream);
        if (st) return st;
    }
    return dabgpu_stream_bank_process(b, b->d_raw_scratch, padded, n_samples, d_bits, max_frames_per_stream, d_n_frames, stream);
}

// frames go straight into per-stream frame-history rings (the layout dabgpu_fic_decode_ring / dabgpu_msc_decode_ring read)
static int bank_process_ring_any(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                 int8_t* d_hist, int hist_frames, int32_t* d_newest_slot, int bits_layout, void* stream, const void* d_prev, int retain) {
    if (!b || !d_raw || !d_hist || !d_newest_slot) { dabgpu_set_error("stream_bank_process_ring: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (hist_frames < 5) { dabgpu_set_error("stream_bank_process_ring: the ring needs at least 5 frames"); return DABGPU_ERR_INVALID_ARG; }
    if (bits_layout != DABGPU_BITS_NATURAL && bits_layout != DABGPU_BITS_MSC_CLASSED) {
        dabgpu_set_error("stream_bank_process_ring: unknown bits_layout %d", bits_layout); return DABGPU_ERR_INVALID_ARG;
    }
    const int classed = bits_layout == DABGPU_BITS_MSC_CLASSED;
    if (classed && b->view.g.mode != 1) {
        dabgpu_set_error("stream_bank_process_ring: class order is defined for transmission mode I (the DAB layer above the soft bits is mode I only)");
        return DABGPU_ERR_UNSUPPORTED;
    }
    switch (format) {
    case DABGPU_IQ_RAW_F32L: case DABGPU_IQ_WAV_F32:
        return bank_process_impl<0>(b, d_raw, stream_stride_samples, n_samples, d_hist, (size_t)hist_frames, d_newest_slot, stream, 1, classed, d_prev, retain);
    case DABGPU_IQ_RAW_U8: case DABGPU_IQ_WAV_PCM8:
        return bank_process_impl<1>(b, d_raw, stream_stride_samples, n_samples, d_hist, (size_t)hist_frames, d_newest_slot, stream, 1, classed, d_prev, retain);
    case DABGPU_IQ_RAW_S8:
        return bank_process_impl<2>(b, d_raw, stream_stride_samples, n_samples, d_hist, (size_t)hist_frames, d_newest_slot, stream, 1, classed, d_prev, retain);
    case DABGPU_IQ_RAW_S16L: case DABGPU_IQ_WAV_PCM16:
        return bank_process_impl<3>(b, d_raw, stream_stride_samples, n_samples, d_hist, (size_t)hist_frames, d_newest_slot, stream, 1, classed, d_prev, retain);
    default:
        dabgpu_set_error("stream_bank_process_ring: format %d is not read directly (use raw_f32l, raw_u8, raw_s8 or raw_s16l)", format);
        return DABGPU_ERR_UNSUPPORTED;
    }
}

int dabgpu_stream_bank_process_ring_layout(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                           int8_t* d_hist, int hist_frames, int32_t* d_newest_slot, int bits_layout, void* stream) {
    return bank_process_ring_any(b, d_raw, format, stream_stride_samples, n_samples, d_hist, hist_frames, d_newest_slot, bits_layout, stream, nullptr, 0);
}

int dabgpu_stream_bank_process_ring_retained(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                             const void* d_prev_raw, int8_t* d_hist, int hist_frames, int32_t* d_newest_slot, int bits_layout,
                                             void* stream) {
    return bank_process_ring_any(b, d_raw, format, stream_stride_samples, n_samples, d_hist, hist_frames, d_newest_slot, bits_layout, stream, d_prev_raw, 1);
}

int dabgpu_stream_bank_process_ring(dabgpu_stream_bank* b, const void* d_raw, int format, size_t stream_stride_samples, size_t n_samples,
                                    int8_t* d_hist, int hist_frames, int32_t* d_newest_slot, void* stream) {
    return dabgpu_stream_bank_process_ring_layout(b, d_raw, format, stream_stride_samples, n_samples, d_hist, hist_frames, d_newest_slot,
                                                  DABGPU_BITS_NATURAL, stream);
}

int dabgpu_stream_bank_status(dabgpu_stream_bank* b, dabgpu_stream_status* h_status, void* stream) {
    if (!b || !h_status) { dabgpu_set_error("stream_bank_status: null argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(b->ctx);
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)b->n;
    int st;
    hipLaunchKernelGGL(stream_report_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, b->view, n, (int*)nullptr, b->d_status, 0);
    if ((st = dabgpu_check_hip(hipGetLastError(), "stream_report_kernel launch"))) return st;
    if ((st = dabgpu_check_hip(hipMemcpyAsync(h_status, b->d_status, b->n * sizeof(dabgpu_stream_status), hipMemcpyDeviceToHost, s), "hipMemcpyAsync"))) return st;
    return dabgpu_check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
}

}  // extern "C"
