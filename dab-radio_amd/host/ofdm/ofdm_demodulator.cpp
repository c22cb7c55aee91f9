// ofdm/ofdm_demodulator.cpp -- see ofdm_demodulator.h.  Reference cited: src/ofdm/ofdm_demodulator.cpp.
#include "./ofdm_demodulator.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "dab/dabgpu_frame_batcher.h"

namespace {
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
    const char* dev = std::getenv("DABGPU_DEVICE");
    int st;
    if (m_mode == 1) {
        st = dabgpu_create(&m_ctx, dev ? std::atoi(dev) : 0, reinterpret_cast<const float*>(prs_fft_ref.data()), carrier_mapper.data());
    } else {
        // modes II-IV run on the library's built-in tables of that mode: a caller-supplied table must be that table
        std::vector<float> prs(2 * params.nb_fft);
        std::vector<int> map(params.nb_data_carriers);
        if (dabgpu_get_prs_fft_ref(m_mode, prs.data()) != DABGPU_OK || dabgpu_get_carrier_mapper(m_mode, map.data()) != DABGPU_OK ||
            std::memcmp(prs.data(), prs_fft_ref.data(), prs.size() * sizeof(float)) != 0 ||
            std::memcmp(map.data(), carrier_mapper.data(), map.size() * sizeof(int)) != 0)
            throw std::runtime_error("OFDM_Demod: custom PRS / carrier tables are only supported in transmission mode I");
        st = dabgpu_create(&m_ctx, dev ? std::atoi(dev) : 0, nullptr, nullptr);
    }
    if (st != DABGPU_OK) fail("dabgpu_create", st);       // no CPU fallback: a missing GPU is a construction error
    m_ring.assign(params.nb_null_period, {0.0f, 0.0f});
    m_corr.assign(params.nb_null_period + params.nb_symbol_period, {0.0f, 0.0f});
    m_frame.assign(params.nb_frame_symbols * params.nb_symbol_period + params.nb_null_period, {0.0f, 0.0f});
    m_frame_bits.assign((params.nb_frame_symbols - 1) * params.nb_data_carriers * 2, 0);
    m_frame_fft.assign((params.nb_frame_symbols + 1) * params.nb_fft, {0.0f, 0.0f});
    m_frame_dqpsk.assign((params.nb_frame_symbols - 1) * params.nb_fft, {0.0f, 0.0f});
    m_impulse_response.assign(params.nb_fft, 0.0f);
    m_frequency_response.assign(params.nb_fft, 0.0f);
    // the frame's samples go to the device and its soft bits come back once per frame: page-locked, the two copies run at PCIe speed
    // (best effort: an unpinned buffer works too)
    m_pinned_frame = dabgpu_host_pin(m_frame.data(), m_frame.size() * sizeof(m_frame[0])) == DABGPU_OK;
    m_pinned_bits = dabgpu_host_pin(m_frame_bits.data(), m_frame_bits.size() * sizeof(m_frame_bits[0])) == DABGPU_OK;
}

OFDM_Demod::~OFDM_Demod() {
    dabgpu_frame_batcher::remove_producer(this);
    if (m_pinned_frame) (void)dabgpu_host_unpin(m_frame.data());
    if (m_pinned_bits) (void)dabgpu_host_unpin(m_frame_bits.data());
    dabgpu_destroy(m_ctx);
}

// :235-275
void OFDM_Demod::Process(tcb::span<const std::complex<float>> buf) {
    UpdateSignalAverage(buf);
    size_t pos = 0;
    while (pos < buf.size()) {
        auto rest = buf.subspan(pos);
        switch (m_state) {
        case FINDING_NULL_POWER_DIP: pos += FindNullPowerDip(rest); break;
        case READING_NULL_AND_PRS: pos += ReadNullPRS(rest); break;
        case RUNNING_COARSE_FREQ_SYNC:
        case RUNNING_FINE_TIME_SYNC: RunSync(); break;
        case READING_SYMBOLS: pos += ReadSymbols(rest); break;
        }
    }
}

// :277-289
void OFDM_Demod::Reset() {
    m_state = FINDING_NULL_POWER_DIP;
    m_corr_length = 0;
    m_total_frames_desync++;
    m_is_found_coarse = false;
    m_freq_coarse = 0.0f;
    m_freq_fine = 0.0f;
    m_fine_time_offset = 0;
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

// :360-471 + :473-548, both on the device in one launch
void OFDM_Demod::RunSync() {
    dabgpu_sync_cfg cfg;
    cfg.fine_freq_update_beta = m_cfg.sync.fine_freq_update_beta;
    cfg.is_coarse_freq_correction = m_cfg.sync.is_coarse_freq_correction ? 1 : 0;
    cfg.max_coarse_freq_correction_norm = m_cfg.sync.max_coarse_freq_correction_norm;
    cfg.coarse_freq_slow_beta = m_cfg.sync.coarse_freq_slow_beta;
    cfg.impulse_peak_threshold_db = m_cfg.sync.impulse_peak_threshold_db;
    cfg.impulse_peak_distance_probability = m_cfg.sync.impulse_peak_distance_probability;
    dabgpu_sync_state st;
    st.freq_coarse = m_freq_coarse;
    st.freq_fine = m_freq_fine;
    st.is_found_coarse = m_is_found_coarse ? 1 : 0;
    st.fine_time_offset = 0;
    st.sync_valid = 0;
    st.reserved = 0;
    m_state = RUNNING_FINE_TIME_SYNC;
    const auto* prs_sym = &m_corr[m_params.nb_null_period];
    const int rc = dabgpu_ofdm_sync_host_sync_mode(m_ctx, m_mode, reinterpret_cast<const float*>(prs_sym), &cfg, &st,
                                                   m_impulse_response.data(),
                                                   cfg.is_coarse_freq_correction ? m_frequency_response.data() : nullptr);
    if (rc != DABGPU_OK) fail("dabgpu_ofdm_sync_host_sync_mode", rc);
    m_freq_coarse = st.freq_coarse;
    m_freq_fine = st.freq_fine;
    m_is_found_coarse = st.is_found_coarse != 0;
    if (!st.sync_valid) { Reset(); return; }                  // :529-532
    // :536-546 the PRS starts `offset` samples after (or before) the expected position
    const int offset = st.fine_time_offset;
    const size_t start = (size_t)((int)m_params.nb_null_period + offset);
    const size_t count = m_corr.size() - start;
    std::copy_n(m_corr.begin() + (std::ptrdiff_t)start, count, m_frame.begin());
    m_frame_length = count;
    m_corr_length = 0;
    m_fine_time_offset = offset;
    m_state = READING_SYMBOLS;
}

// :550-577
size_t OFDM_Demod::ReadSymbols(tcb::span<const std::complex<float>> buf) {
    const size_t want = m_frame.size() - m_frame_length;
    const size_t take = std::min(want, buf.size());
    std::copy_n(buf.begin(), take, m_frame.begin() + (std::ptrdiff_t)m_frame_length);
    m_frame_length += take;
    if (m_frame_length < m_frame.size()) return take;
    // the NULL symbol at the end of this frame heads the next correlation window (:558-562)
    const size_t null_at = m_params.nb_frame_symbols * m_params.nb_symbol_period;
    std::copy_n(m_frame.begin() + (std::ptrdiff_t)null_at, m_params.nb_null_period, m_corr.begin());
    m_corr_length = m_params.nb_null_period;
    DemodulateFrame();
    m_frame_length = 0;
    m_state = READING_NULL_AND_PRS;
    return take;
}

// :581-639 (coordinator) + :650-766 (pipelines) for one frame
void OFDM_Demod::DemodulateFrame() {
    float fine = m_freq_fine;
    // mode I: the register-resident kernel (also fills GetFrameDataVec); modes II-IV: the size-generic kernel
    const int rc = (m_mode == 1)
        ? dabgpu_ofdm_demod_stream_frame_sync(
              m_ctx, reinterpret_cast<const float*>(m_frame.data()), m_freq_coarse, &fine, m_cfg.sync.fine_freq_update_beta,
              m_frame_bits.data(), nullptr, m_fetch_debug ? reinterpret_cast<float*>(m_frame_fft.data()) : nullptr,
              m_fetch_debug ? reinterpret_cast<float*>(m_frame_dqpsk.data()) : nullptr)
        : dabgpu_ofdm_demod_stream_frame_sync_mode(
              m_ctx, m_mode, reinterpret_cast<const float*>(m_frame.data()), m_freq_coarse, &fine, m_cfg.sync.fine_freq_update_beta,
              m_frame_bits.data(), nullptr, m_fetch_debug ? reinterpret_cast<float*>(m_frame_fft.data()) : nullptr);
    if (rc != DABGPU_OK) fail("dabgpu_ofdm_demod_stream_frame_sync", rc);
    m_freq_fine = fine;
    m_total_frames_read++;
    // the decoders of this process get the whole frame decoded in one go (mode I: the DAB layer above the soft bits is mode I only)
    if (m_mode == 1) dabgpu_frame_batcher::on_frame(this, m_frame_bits.data());
    m_on_frame.Notify(tcb::span<const viterbi_bit_t>(m_frame_bits.data(), m_frame_bits.size()));
}
