// ofdm/ofdm_demodulator.cpp -- see ofdm_demodulator.h.  Reference cited: src/ofdm/ofdm_demodulator.cpp.
#include "./ofdm_demodulator.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include <pthread.h>

#include "dabgpu.h"
#include "dab/dabgpu_frame_batcher.h"
#include "dab/dabgpu_shared_context.h"

namespace {
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
[[noreturn]] void fail(const char* what, int st) {
    throw std::runtime_error(std::string("OFDM_Demod: ") + what + ": " + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
}
}  // namespace

OFDM_Demod::OFDM_Demod(const OFDM_Params& params, const tcb::span<const std::complex<float>> prs_fft_ref,
                       const tcb::span<const int> carrier_mapper, int /*nb_desired_threads*/)
    : m_params(params) {
    // the kernels exist for the four DAB geometries (src/ofdm/dab_ofdm_params_ref.cpp:11-60); anything else has no device path
    for (int mode = 1; mode <= 4 && m_mode == 0; mode++) {
        int g[9];
        if (dabgpu_get_ofdm_params(mode, g) != DABGPU_OK) continue;
        if ((int)params.nb_frame_symbols == g[0] && (int)params.nb_symbol_period == g[1] && (int)params.nb_null_period == g[2] &&
            (int)params.nb_fft == g[3] && (int)params.nb_cyclic_prefix == g[4] && (int)params.nb_data_carriers == g[5]) m_mode = mode;
    }
    if (m_mode == 0) throw std::runtime_error("OFDM_Demod: the MI355X kernels implement the DAB transmission modes I-IV only");
    if (prs_fft_ref.size() < params.nb_fft || carrier_mapper.size() < params.nb_data_carriers)
        throw std::runtime_error("OFDM_Demod: PRS reference / carrier mapper too small");
    if (dabgpu_abi_version() != DABGPU_ABI_VERSION)
        throw std::runtime_error("OFDM_Demod: libdabgpu.so implements ABI version " + std::to_string(dabgpu_abi_version()) + ", these classes were built for " +
                                 std::to_string(DABGPU_ABI_VERSION));
    const char* dev = std::getenv("DABGPU_DEVICE");
    int st;
    if (m_mode == 1) {
        // Many receivers in one process share ONE pair of device streams (the receiver bank, csrc/receiver_bank.hip): what they post is issued as
        // one demodulation launch, one synchroniser launch and one decode over all of them.  Measured in the steady state (tools/bench_mirror_multi.py, 2000
        // frames per receiver): up to four receivers are faster on pipelines of their own (9-11 k frames/s in total at 2-4 against 5-8 k), at
        // eight the bank is level or ahead, from sixteen on it is 2.5-5 x faster (12-17 k against 3-5 k) -- and a member costs 6 MB of host memory where a
        // pipeline costs 230.  DABGPU_MIRROR_BANK=1: every mode I receiver on the library's own tables joins the bank; 0: none; unset: a receiver joins when
        // DABGPU_MIRROR_BANK_FROM - 1 (default 1) others are alive at its construction -- the first receiver of a process keeps a pipeline of its own (one
        // receiver is the common case and the fastest on it), every further one is a member: 7 k frames/s at 2-4 receivers instead of 9-11 k (a live signal
        // needs 10.4 per receiver), 9-13 k at 8-32 instead of 7.5-12 k with four private pipelines beside the bank, and 230 MB less host memory for each.
        static std::atomic<int> live{0};
        const char* bank_env = std::getenv("DABGPU_MIRROR_BANK");
        const char* from_env = std::getenv("DABGPU_MIRROR_BANK_FROM");
        const int from = from_env ? std::max(1, std::atoi(from_env)) : 2;
        const int others = live.fetch_add(1);
        m_counted = true;
        bool banked = bank_env ? std::atoi(bank_env) != 0 : others >= from - 1;
        if (banked) {
            std::vector<float> prs(2 * params.nb_fft);
            std::vector<int> map(params.nb_data_carriers);
            banked = dabgpu_get_prs_fft_ref(1, prs.data()) == DABGPU_OK && dabgpu_get_carrier_mapper(1, map.data()) == DABGPU_OK &&
                     std::memcmp(prs.data(), prs_fft_ref.data(), prs.size() * sizeof(float)) == 0 &&
                     std::memcmp(map.data(), carrier_mapper.data(), map.size() * sizeof(int)) == 0;
        }
        m_live = &live;
        m_banked = banked;
        st = DABGPU_OK;
        if (banked) {
            st = dabgpu_receiver_create_banked(&m_rx, dev ? std::atoi(dev) : 0);
            // (a full bank -- 64 members per device -- is no reason to fail when nobody asked for the bank: the receiver gets a pipeline of its own)
            if (st != DABGPU_OK && !bank_env) { m_banked = banked = false; m_rx = nullptr; }
        }
        if (!banked) st = dabgpu_receiver_create(&m_rx, dev ? std::atoi(dev) : 0, 1, reinterpret_cast<const float*>(prs_fft_ref.data()), carrier_mapper.data());
        if (st != DABGPU_OK) live.fetch_sub(1);
    } else {
        // modes II-IV run on the library's built-in tables of that mode: a caller-supplied table must be that table
        std::vector<float> prs(2 * params.nb_fft);
        std::vector<int> map(params.nb_data_carriers);
        if (dabgpu_get_prs_fft_ref(m_mode, prs.data()) != DABGPU_OK || dabgpu_get_carrier_mapper(m_mode, map.data()) != DABGPU_OK ||
            std::memcmp(prs.data(), prs_fft_ref.data(), prs.size() * sizeof(float)) != 0 ||
            std::memcmp(map.data(), carrier_mapper.data(), map.size() * sizeof(int)) != 0)
            throw std::runtime_error("OFDM_Demod: custom PRS / carrier tables are only supported in transmission mode I");
        st = dabgpu_receiver_create(&m_rx, dev ? std::atoi(dev) : 0, m_mode, nullptr, nullptr);
    }
    if (st != DABGPU_OK) fail("dabgpu_receiver_create", st);       // no CPU fallback: a missing GPU is a construction error
    float* stage = nullptr;
    if ((st = dabgpu_receiver_stage(m_rx, &stage, &m_stage_capacity)) != DABGPU_OK) { dabgpu_receiver_destroy(m_rx); fail("dabgpu_receiver_stage", st); }
    m_stage = reinterpret_cast<std::complex<float>*>(stage);
    if (const char* d = std::getenv("DABGPU_MIRROR_DEPTH")) m_depth = std::min(6, std::max(1, std::atoi(d)));
    if (const char* d = std::getenv("DABGPU_MIRROR_PROFILE")) m_profile = std::atoi(d) != 0;
    m_ring.assign(params.nb_null_period, {0.0f, 0.0f});
    m_corr.assign(params.nb_null_period + params.nb_symbol_period, {0.0f, 0.0f});
    m_frame_bits.assign((params.nb_frame_symbols - 1) * params.nb_data_carriers * 2, 0);
    m_bits_ptr.store(m_frame_bits.data());
    m_bits_len = m_frame_bits.size();
    m_frame_fft.assign((params.nb_frame_symbols + 1) * params.nb_fft, {0.0f, 0.0f});
    m_frame_dqpsk.assign((params.nb_frame_symbols - 1) * params.nb_fft, {0.0f, 0.0f});
    m_impulse_response.assign(params.nb_fft, 0.0f);
    m_frequency_response.assign(params.nb_fft, 0.0f);
    m_thread = std::thread([this] { DeliveryThread(); });
    m_decode_thread = std::thread([this] { DecodeThread(); });
}

OFDM_Demod::~OFDM_Demod() {
    {
        // every frame handed to the device is delivered before the object goes (the reference's destructor lets the coordinator finish too)
        std::unique_lock<std::mutex> lock(m_mu);
        m_cv_done.wait(lock, [this] { return m_items.empty() && !m_busy; });
        m_stop = true;
    }
    m_cv_items.notify_all();
    m_cv_decode.notify_all();
    if (m_thread.joinable()) m_thread.join();
    if (m_decode_thread.joinable()) m_decode_thread.join();
    dabgpu_frame_batcher::remove_producer(this);
    dabgpu_receiver_destroy(m_rx);
    if (m_counted && m_live) m_live->fetch_sub(1);
    if (m_profile && m_total_frames_read.load() > 0) {
        const double n = (double)m_total_frames_read.load();
        std::fprintf(stderr, "OFDM_Demod profile, us per frame over %d frames: reader { wait for the synchroniser %.1f, wait for a slot %.1f, submit %.1f } "
                             "delivery { wait for the frame %.1f, batcher %.1f, observers %.1f }\n", m_total_frames_read.load(), m_t_sync_wait / n,
                     m_t_slot_wait / n, m_t_submit / n, m_t_frame_wait / n, m_t_batcher / n, m_t_observers / n);
    }
}

void OFDM_Demod::RethrowDeliveryError() {
    std::exception_ptr e;
    { std::lock_guard<std::mutex> lock(m_mu); e = m_error; m_error = nullptr; }
    if (e) std::rethrow_exception(e);
}

// not in the reference: the point at which everything Process() was handed has come out of the observers
void OFDM_Demod::Synchronize() {
    CollectPendingSync();                     // (the state the getters show is the one the serial machine would be in after the samples handed in)
    {
        std::unique_lock<std::mutex> lock(m_mu);
        m_cv_done.wait(lock, [this] { return m_items.empty() && !m_busy; });
    }
    RethrowDeliveryError();
}

// :235-275
void OFDM_Demod::Process(tcb::span<const std::complex<float>> buf) {
    RethrowDeliveryError();
    UpdateSignalAverage(buf);
    // A synchroniser submitted in an earlier block is NOT waited for here: the reader buffers on until the earliest sample at which the frame can end
    // (Run) and collects the record there -- the device's answer travels while the rest of the frame is copied.  Should the impulse-peak test fail, the
    // samples buffered behind the PRS slot go through the NULL search they would have gone through (:529-532 Reset, then FindNullPowerDip ...), each
    // with the signal average of the Process() call that delivered it: the calls' boundaries inside the staging buffer are noted for that.
    // (one stretch per call, never merged: FindNullPowerDip's windows are aligned to the span it is handed, :291-347)
    if (m_sync_pending) m_spec_segments.push_back(Segment{m_stage_length, m_signal_l1_average});
    Run(buf);
}

// blocking: Synchronize(), Reset()
void OFDM_Demod::CollectPendingSync() {
    if (!m_sync_pending) return;
    const size_t buffered = m_stage_length;
    if (!ResolveSync()) ReplayAfterFailedSync(buffered);
}

// The impulse-peak test of the synchroniser submitted last failed (ResolveSync has reset the reader: one more desync): the serial machine knew that
// when the PRS slot was complete and went on searching from the next sample -- do that now with what was buffered meanwhile, serially (a
// synchroniser submitted on the way is waited for at once), every stretch under the signal average of the call that delivered it.
void OFDM_Demod::ReplayAfterFailedSync(size_t buffered) {
    const size_t head = m_corr.size();
    const std::vector<std::complex<float>> rest(m_stage + head, m_stage + buffered);
    const std::vector<Segment> segments = std::move(m_spec_segments);
    m_spec_segments.clear();
    const float average_now = m_signal_l1_average;
    m_replaying = true;
    for (size_t k = 0; k < segments.size(); k++) {
        const size_t begin = std::min(segments[k].start, buffered) - head, end = std::min(k + 1 < segments.size() ? segments[k + 1].start : buffered, buffered) - head;
        if (end <= begin) continue;
        m_signal_l1_average = segments[k].average;
        Run(tcb::span<const std::complex<float>>(rest.data() + begin, end - begin));
    }
    m_signal_l1_average = average_now;
    m_replaying = false;
}

void OFDM_Demod::Run(tcb::span<const std::complex<float>> buf) {
    size_t pos = 0;
    // where this span's samples behind a PRS slot begin: in the span, and in the staging buffer (a record outstanding from an earlier call: 0 / what was
    // buffered before this call)
    size_t spec_pos = 0, spec_stage = m_stage_length;
    while (pos < buf.size()) {
        auto rest = buf.subspan(pos);
        switch (m_state) {
        case FINDING_NULL_POWER_DIP: pos += FindNullPowerDip(rest); break;
        case READING_NULL_AND_PRS:
            pos += ReadNullPRS(rest);
            if (m_state == RUNNING_COARSE_FREQ_SYNC) {
                SubmitSync();
                spec_pos = pos; spec_stage = m_stage_length;
                if (m_replaying) (void)ResolveSync();               // (a replay is serial: false = the reader was reset again, the search goes on)
            }
            break;
        case RUNNING_COARSE_FREQ_SYNC:
        case RUNNING_FINE_TIME_SYNC: {
            // the synchroniser runs on the device; whatever it finds, the samples up to where the earliest possible frame ends belong
            // to this frame if the impulse-peak test passes: buffer on instead of waiting
            const size_t earliest_end = m_params.nb_null_period - m_params.nb_cyclic_prefix +
                                        m_params.nb_frame_symbols * m_params.nb_symbol_period + m_params.nb_null_period;
            const size_t take = std::min(earliest_end - m_stage_length, rest.size());
            std::copy_n(rest.begin(), take, m_stage + m_stage_length);
            m_stage_length += take;
            pos += take;
            if (m_stage_length == earliest_end && !ResolveSync()) {
                // what earlier calls delivered is replayed call by call; this span's own samples are gone through again from where they began behind
                // the PRS slot, as ONE span (the serial machine never saw them split at the point where this reader happened to stop buffering)
                ReplayAfterFailedSync(spec_stage);
                pos = spec_pos;
            }
            break;
        }
        case READING_SYMBOLS: pos += ReadSymbols(rest); break;
        }
    }
}

// :277-289
void OFDM_Demod::Reset() {
    // a record still in flight is collected first: the serial machine had its answer when the PRS slot was complete -- had the impulse-peak test
    // failed, it had already reset itself (one more desync) and searched the rest of that block for the NULL symbol before this call
    CollectPendingSync();
    ResetReader();
}

void OFDM_Demod::ResetReader() {
    m_state = FINDING_NULL_POWER_DIP;
    m_corr_length = 0;
    m_stage_length = 0;
    const int rc = dabgpu_receiver_reset(m_rx);       // coarse = fine = 0, no coarse offset found -- behind the frames already submitted
    if (rc != DABGPU_OK) fail("dabgpu_receiver_reset", rc);
    {
        std::lock_guard<std::mutex> lock(m_mu);
        m_items.push_back(Item{Item::RESET, 0.0f, 0.0f, 0, 0, false, false});
    }
    m_cv_items.notify_one();
}

// :922-932
float OFDM_Demod::L1Average(const std::complex<float>* block, size_t n) const {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) acc += std::abs(block[i].real()) + std::abs(block[i].imag());
    return acc / (float)n;
}

// :934-950
void OFDM_Demod::UpdateSignalAverage(tcb::span<const std::complex<float>> block) {
    const size_t n = block.size(), k = (size_t)m_cfg.signal_l1.nb_samples;
    if (n < k) return;
    const size_t stride = k * (size_t)m_cfg.signal_l1.nb_decimate;
    const float beta = m_cfg.signal_l1.update_beta;
    for (size_t i = 0; i < n - k; i += stride)
        m_signal_l1_average = beta * m_signal_l1_average + (1.0f - beta) * L1Average(&block[i], k);
}

// :291-347
size_t OFDM_Demod::FindNullPowerDip(tcb::span<const std::complex<float>> buf) {
    const int n = (int)buf.size(), k = m_cfg.signal_l1.nb_samples;
    const float start_thresh = m_signal_l1_average * m_cfg.null_l1_search.thresh_null_start;
    const float end_thresh = m_signal_l1_average * m_cfg.null_l1_search.thresh_null_end;
    int nb_read = n;
    for (int i = 0; i < n - k; i += k) {
        const float l1 = L1Average(&buf[(size_t)i], (size_t)k);
        if (m_null_start_found) {
            if (l1 > end_thresh) { m_null_end_found = true; nb_read = i + k; break; }
        } else if (l1 < start_thresh) {
            m_null_start_found = true;
        }
    }
    // keep the last nb_null_period samples (:325, circular_buffer.h:18-37 with read_all)
    const size_t cap = m_ring.size();
    for (int i = 0; i < nb_read; i++) { m_ring[m_ring_index] = buf[(size_t)i]; m_ring_index = (m_ring_index + 1) % cap; }
    m_ring_length = std::min(cap, m_ring_length + (size_t)nb_read);
    if (!m_null_end_found) return (size_t)nb_read;
    // the captured NULL becomes the head of the correlation window (:333-338)
    for (size_t i = 0; i < m_ring_length; i++) m_corr[i] = m_ring[(i + m_ring_index) % cap];
    m_corr_length = m_ring_length;
    m_null_start_found = false;
    m_null_end_found = false;
    m_ring_length = 0;
    m_state = READING_NULL_AND_PRS;
    return (size_t)nb_read;
}

// :349-358
size_t OFDM_Demod::ReadNullPRS(tcb::span<const std::complex<float>> buf) {
    const size_t want = m_corr.size() - m_corr_length;
    const size_t take = std::min(want, buf.size());
    std::copy_n(buf.begin(), take, m_corr.begin() + (std::ptrdiff_t)m_corr_length);
    m_corr_length += take;
    if (m_corr_length == m_corr.size()) m_state = RUNNING_COARSE_FREQ_SYNC;
    return take;
}

// :360-471 + :473-548 on the device, one launch, enqueued: the correlation window heads the staging buffer the frame is assembled in
void OFDM_Demod::SubmitSync() {
    dabgpu_sync_cfg cfg;
    cfg.fine_freq_update_beta = m_cfg.sync.fine_freq_update_beta;
    cfg.is_coarse_freq_correction = m_cfg.sync.is_coarse_freq_correction ? 1 : 0;
    cfg.max_coarse_freq_correction_norm = m_cfg.sync.max_coarse_freq_correction_norm;
    cfg.coarse_freq_slow_beta = m_cfg.sync.coarse_freq_slow_beta;
    cfg.impulse_peak_threshold_db = m_cfg.sync.impulse_peak_threshold_db;
    cfg.impulse_peak_distance_probability = m_cfg.sync.impulse_peak_distance_probability;
    std::copy(m_corr.begin(), m_corr.end(), m_stage);
    m_stage_length = m_corr.size();
    m_state = RUNNING_FINE_TIME_SYNC;
    m_spec_segments.assign(1, Segment{m_stage_length, m_signal_l1_average});
    const int rc = dabgpu_receiver_submit_sync(m_rx, &cfg, m_params.nb_null_period);
    if (rc != DABGPU_OK) fail("dabgpu_receiver_submit_sync", rc);
    m_sync_pending = true;
}

// the record of the synchroniser submitted last; false = the impulse-peak test failed and the receiver was reset (:529-532)
bool OFDM_Demod::ResolveSync() {
    dabgpu_sync_state st;
    const double t0 = m_profile ? now_us() : 0.0;
    const int rc = dabgpu_receiver_wait_sync(m_rx, &st, m_impulse_response.data(),
                                             m_cfg.sync.is_coarse_freq_correction ? m_frequency_response.data() : nullptr);
    if (m_profile) m_t_sync_wait += now_us() - t0;
    m_sync_pending = false;
    if (rc != DABGPU_OK) fail("dabgpu_receiver_wait_sync", rc);
    {
        std::lock_guard<std::mutex> lock(m_mu);
        m_items.push_back(Item{Item::SYNC, st.freq_coarse, st.freq_fine, st.sync_valid ? st.fine_time_offset : 0, 0, false, false});
    }
    m_cv_items.notify_one();
    m_corr_length = 0;
    if (!st.sync_valid) { ResetReader(); return false; }
    // :536-546 the PRS starts `offset` samples after (or before) the expected position
    m_reader_time_offset = st.fine_time_offset;
    m_frame_end = (size_t)((int)m_params.nb_null_period + st.fine_time_offset) + m_params.nb_frame_symbols * m_params.nb_symbol_period + m_params.nb_null_period;
    m_state = READING_SYMBOLS;
    return true;
}

// :550-577
size_t OFDM_Demod::ReadSymbols(tcb::span<const std::complex<float>> buf) {
    const size_t want = m_frame_end - m_stage_length;
    const size_t take = std::min(want, buf.size());
    std::copy_n(buf.begin(), take, m_stage + m_stage_length);
    m_stage_length += take;
    if (m_stage_length < m_frame_end) return take;
    // the NULL symbol at the end of this frame heads the next correlation window (:558-562)
    std::copy_n(m_stage + (m_frame_end - m_params.nb_null_period), m_params.nb_null_period, m_corr.begin());
    m_corr_length = m_params.nb_null_period;
    SubmitFrame();
    m_stage_length = 0;
    m_state = READING_NULL_AND_PRS;
    return take;
}

// :565-572 wait for a free slot, swap buffers, start the workers: upload, demodulation, fine-frequency update and -- for the decoders of
// this process that listen -- the frame's FIC / MSC decode, enqueued; the delivery thread picks the results up
void OFDM_Demod::SubmitFrame() {
    const double t0 = m_profile ? now_us() : 0.0;
    {
        std::unique_lock<std::mutex> lock(m_mu);
        m_cv_done.wait(lock, [this] { return m_frames_in_flight < m_depth; });
        // (the demodulation of frame g overwrites what the decode of frame g - 4 reads: never more than 4 frames ahead of the decode thread)
        m_cv_decoded.wait(lock, [this] { return m_frames_submitted < m_decodes_submitted + 4 || (bool)m_error; });
    }
    const double t1 = m_profile ? now_us() : 0.0;
    // what the decoders of this process listen to NOW -- asked after the wait for a slot: an observer that created a decoder while this thread
    // waited is heard one frame earlier.  The list in force when a frame's DECODE is enqueued applies (the decode thread may be up to 4 frames
    // behind this one: those frames, demodulated already, are decoded with the new list too -- every result slot records the list it was decoded
    // with, and a decoder only picks up what was decoded for it, dabgpu_frame_session_fetch_cif).
    if (m_mode == 1) {
        std::vector<dabgpu_subchannel> subs;
        bool fic = false;
        const uint64_t version = dabgpu_frame_batcher::subscription(subs, fic);
        if (version != m_subs_version) {
            int rc = dabgpu_receiver_set_subchannels(m_rx, subs.data(), (int)subs.size(), fic ? 1 : 0);
            if (rc != DABGPU_OK) {
                // a list the session cannot take (more code words than its scratch may grow to, ...): the stream goes on, the frames are not decoded
                // with it and the decoders work call by call -- as they do for every frame submitted before they existed
                std::fprintf(stderr, "OFDM_Demod: the frame session refused the decoders' sub-channel list (%s -- %s): decoding call by call\n", dabgpu_strerror(rc),
                             dabgpu_last_error());
                rc = dabgpu_receiver_set_subchannels(m_rx, nullptr, 0, 0);
                if (rc != DABGPU_OK) fail("dabgpu_receiver_set_subchannels", rc);
            }
            m_subs_version = version;
        }
    }
    RethrowDeliveryError();
    uint64_t gen = 0;
    const bool fetch_debug = m_fetch_debug.load(std::memory_order_relaxed);
    const size_t frame_sample = (size_t)((int)m_params.nb_null_period + m_reader_time_offset);
    // this thread enqueues the frame's upload, demodulation and fine-frequency update; the decode thread waits for the demodulation and enqueues the
    // decode (dabgpu_receiver_submit_demod / _submit_decode: no stream waits for another on the device, and half the runtime calls leave this thread)
    // (a member of the receiver bank posts the frame in one call: the bank's round demodulates and decodes it)
    const int rc = m_banked ? dabgpu_receiver_submit_frame(m_rx, frame_sample, m_cfg.sync.fine_freq_update_beta, fetch_debug ? 1 : 0, dabgpu_tie_rule_from_env(), &gen)
                            : dabgpu_receiver_submit_demod(m_rx, frame_sample, m_cfg.sync.fine_freq_update_beta, fetch_debug ? 1 : 0, &gen);
    if (rc != DABGPU_OK) fail(m_banked ? "dabgpu_receiver_submit_frame" : "dabgpu_receiver_submit_demod", rc);
    float* stage = nullptr;
    (void)dabgpu_receiver_stage(m_rx, &stage, nullptr);
    m_stage = reinterpret_cast<std::complex<float>*>(stage);
    if (m_profile) { m_t_slot_wait += t1 - t0; m_t_submit += now_us() - t1; }
    {
        std::lock_guard<std::mutex> lock(m_mu);
        m_items.push_back(Item{Item::FRAME, 0.0f, 0.0f, 0, gen, fetch_debug, m_mode == 1});
        m_frames_in_flight++;
        m_frames_submitted++;
        if (m_banked) m_decodes_submitted = gen + 1;            // (nothing is left for the decode thread)
        else m_to_decode.push_back(gen);
    }
    m_cv_items.notify_one();
    m_cv_decode.notify_one();
}

// the generations in submission order: wait (on the host) until the frame is demodulated, enqueue its decode
void OFDM_Demod::DecodeThread() {
    pthread_setname_np(pthread_self(), "dabgpu-decode");
    for (;;) {
        uint64_t gen;
        {
            std::unique_lock<std::mutex> lock(m_mu);
            m_cv_decode.wait(lock, [this] { return m_stop || !m_to_decode.empty(); });
            if (m_to_decode.empty()) return;
            gen = m_to_decode.front();
            m_to_decode.pop_front();
        }
        try {
            const int rc = dabgpu_receiver_submit_decode(m_rx, gen, dabgpu_tie_rule_from_env());
            if (rc != DABGPU_OK) fail("dabgpu_receiver_submit_decode", rc);
        } catch (...) {
            std::lock_guard<std::mutex> lock(m_mu);
            if (!m_error) m_error = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lock(m_mu);
            m_decodes_submitted = gen + 1;
        }
        m_cv_decoded.notify_all();
    }
}

// :581-639 the coordinator's role: everything a frame publishes, in submission order
void OFDM_Demod::DeliveryThread() {
    pthread_setname_np(pthread_self(), "dabgpu-deliver");
    for (;;) {
        Item it;
        {
            std::unique_lock<std::mutex> lock(m_mu);
            m_cv_items.wait(lock, [this] { return m_stop || !m_items.empty(); });
            if (m_items.empty()) return;
            it = m_items.front();
            m_items.pop_front();
            m_busy = true;
        }
        bool failed;
        { std::lock_guard<std::mutex> lock(m_mu); failed = (bool)m_error; }
        try {
            switch (it.kind) {
            case Item::SYNC:
                m_freq_coarse = it.coarse;
                m_freq_fine = it.fine;
                m_fine_time_offset = it.offset;
                break;
            case Item::RESET:
                m_total_frames_desync.fetch_add(1, std::memory_order_relaxed);
                m_freq_coarse = 0.0f;
                m_freq_fine = 0.0f;
                m_fine_time_offset = 0;
                break;
            case Item::FRAME: {
                if (failed) break;                       // after a device error nothing more is delivered; Process() rethrows it
                dabgpu_receiver_frame fr;
                const double t0 = m_profile ? now_us() : 0.0;
                {
                    std::unique_lock<std::mutex> lock(m_mu);                     // (the decode thread has enqueued this frame's decode)
                    m_cv_decoded.wait(lock, [&] { return m_decodes_submitted > it.gen || (bool)m_error; });
                    if (m_error) break;
                }
                const int rc = dabgpu_receiver_wait_frame(m_rx, it.gen, &fr);
                if (rc != DABGPU_OK) fail("dabgpu_receiver_wait_frame", rc);
                const double t1 = m_profile ? now_us() : 0.0;
                m_freq_fine = fr.freq_fine;
                m_total_frames_read.fetch_add(1, std::memory_order_relaxed);
                m_bits_ptr.store(fr.bits, std::memory_order_release);          // (fr.n_bits == m_bits_len: the frame length of the mode)
                if (it.views) {
                    std::memcpy(static_cast<void*>(m_frame_fft.data()), fr.fft, m_frame_fft.size() * sizeof(m_frame_fft[0]));
                    if (m_mode == 1) std::memcpy(static_cast<void*>(m_frame_dqpsk.data()), fr.dqpsk, (m_params.nb_frame_symbols - 1) * m_params.nb_data_carriers * sizeof(m_frame_dqpsk[0]));
                }
                // the decoders of this process find the frame's FIBs and sub-channel bytes already decoded (mode I: the DAB layer above the
                // soft bits is mode I only).  Every frame is shown to the batcher while somebody listens, also one submitted before the
                // listener existed: it has no results to pick up, but a new MSC_Decoder counts its 16 consecutive CIFs from it
                if (it.batch) dabgpu_frame_batcher::on_frame_decoded(this, dabgpu_receiver_session(m_rx), it.gen, fr.bits);
                const double t2 = m_profile ? now_us() : 0.0;
                m_on_frame.Notify(tcb::span<const viterbi_bit_t>(fr.bits, fr.n_bits));
                if (m_profile) { m_t_frame_wait += t1 - t0; m_t_batcher += t2 - t1; m_t_observers += now_us() - t2; }
                break;
            }
            }
        } catch (...) {
            std::lock_guard<std::mutex> lock(m_mu);
            if (!m_error) m_error = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lock(m_mu);
            if (it.kind == Item::FRAME) m_frames_in_flight--;
            m_busy = false;
        }
        m_cv_done.notify_all();
    }
}
