// ofdm/ofdm_demodulator.h -- OFDM_Demod with the reference's public interface (src/ofdm/ofdm_demodulator.h:24-167),
// re-implemented over the MI355X C ABI (include/dabgpu.h).
//
// What runs where:
//   host  : the 5-state framing machine (Process, :235-275), sample buffering, the L1 power statistics used only to
//           find the first NULL symbol (:291-347, :922-950) -- control flow over a few scalars per 100 samples
//   device: coarse frequency sync + fine time sync (one launch per frame, dabgpu_ofdm_sync_host_sync),
//           PLL + cyclic-prefix phase + 77 x FFT + DQPSK + de-interleave + soft bits + fine-frequency loop
//           (one launch per frame, dabgpu_ofdm_demod_stream_frame_sync)
// Ordering contract (replaces the reader / coordinator / pipeline threads of ofdm_demodulator_threads.h): per frame,
// coarse sync -> fine time sync -> demodulation -> fine-frequency update -> On_OFDM_Frame observers, all before the
// next frame's sync; observers are called on the thread that calls Process().
// nb_desired_threads is accepted for source compatibility and ignored (the device kernel replaces the thread pool).
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <complex>
#include <vector>
#include "utility/observable.h"
#include "utility/span.h"
#include "viterbi_config.h"
#include "./ofdm_params.h"

struct dabgpu_ctx;

struct OFDM_Demod_Config {
    struct {
        float update_beta = 0.95f;
        int nb_samples = 100;
        int nb_decimate = 5;
    } signal_l1;
    struct {
        float thresh_null_start = 0.35f;
        float thresh_null_end = 0.75f;
    } null_l1_search;
    struct {
        float fine_freq_update_beta = 0.9f;
        bool is_coarse_freq_correction = true;
        float max_coarse_freq_correction_norm = 0.5f;
        float coarse_freq_slow_beta = 0.1f;
        float impulse_peak_threshold_db = 20.0f;
        float impulse_peak_distance_probability = 0.15f;
    } sync;
};

class OFDM_Demod {
public:
    enum State {
        FINDING_NULL_POWER_DIP,
        READING_NULL_AND_PRS,
        RUNNING_COARSE_FREQ_SYNC,
        RUNNING_FINE_TIME_SYNC,
        READING_SYMBOLS,
    };

    OFDM_Demod(const OFDM_Params& params, const tcb::span<const std::complex<float>> prs_fft_ref,
               const tcb::span<const int> carrier_mapper, int nb_desired_threads = 0);
    ~OFDM_Demod();
    OFDM_Demod(OFDM_Demod&) = delete;
    OFDM_Demod(OFDM_Demod&&) = delete;
    OFDM_Demod& operator=(OFDM_Demod&) = delete;
    OFDM_Demod& operator=(OFDM_Demod&&) = delete;

    void Process(tcb::span<const std::complex<float>> block);
    void Reset();

    OFDM_Params GetOFDMParams() const { return m_params; }
    State GetState() const { return m_state; }
    auto& GetConfig() { return m_cfg; }
    const auto& GetConfig() const { return m_cfg; }
    float GetSignalAverage() const { return m_signal_l1_average; }
    float GetFineFrequencyOffset() const { return m_freq_fine; }
    float GetCoarseFrequencyOffset() const { return m_freq_coarse; }
    float GetNetFrequencyOffset() const { return m_freq_fine + m_freq_coarse; }
    int GetFineTimeOffset() const { return m_fine_time_offset; }
    int GetTotalFramesRead() const { return m_total_frames_read; }
    int GetTotalFramesDesync() const { return m_total_frames_desync; }
    tcb::span<const std::complex<float>> GetFrameFFT() const { return m_frame_fft; }
    tcb::span<const std::complex<float>> GetFrameDataVec() const { return m_frame_dqpsk; }
    tcb::span<const viterbi_bit_t> GetFrameDataBits() const { return m_frame_bits; }
    tcb::span<const float> GetImpulseResponse() const { return m_impulse_response; }
    tcb::span<const float> GetCoarseFrequencyResponse() const { return m_frequency_response; }
    tcb::span<const std::complex<float>> GetCorrelationTimeBuffer() const { return m_corr; }
    auto& On_OFDM_Frame() { return m_on_frame; }
    // GUI buffers (FFT of all 77 symbols, per-carrier DQPSK vectors) cost two extra device->host copies per frame;
    // off by default, switch on before Process() when a viewer reads GetFrameFFT()/GetFrameDataVec()
    void EnableDebugBuffers(bool enable) { m_fetch_debug = enable; }

private:
    size_t FindNullPowerDip(tcb::span<const std::complex<float>> buf);
    size_t ReadNullPRS(tcb::span<const std::complex<float>> buf);
    void RunSync();
    size_t ReadSymbols(tcb::span<const std::complex<float>> buf);
    void DemodulateFrame();
    float L1Average(const std::complex<float>* block, size_t n) const;
    void UpdateSignalAverage(tcb::span<const std::complex<float>> block);

    OFDM_Demod_Config m_cfg;
    State m_state = FINDING_NULL_POWER_DIP;
    const OFDM_Params m_params;
    int m_mode = 0;                          // transmission mode 1..4 matching m_params
    dabgpu_ctx* m_ctx = nullptr;
    int m_total_frames_read = 0;
    int m_total_frames_desync = 0;
    bool m_is_found_coarse = false;
    float m_freq_coarse = 0.0f;
    float m_freq_fine = 0.0f;
    int m_fine_time_offset = 0;
    bool m_null_start_found = false;
    bool m_null_end_found = false;
    float m_signal_l1_average = 0.0f;
    bool m_fetch_debug = false;
    // null search ring (nb_null_period), NULL+PRS correlation window, frame under assembly (76 symbols + NULL)
    std::vector<std::complex<float>> m_ring;
    size_t m_ring_index = 0, m_ring_length = 0;
    std::vector<std::complex<float>> m_corr;
    size_t m_corr_length = 0;
    std::vector<std::complex<float>> m_frame;
    size_t m_frame_length = 0;
    // results
    std::vector<viterbi_bit_t> m_frame_bits;
    bool m_pinned_frame = false, m_pinned_bits = false;      // m_frame / m_frame_bits page-locked (dabgpu_host_pin)
    std::vector<std::complex<float>> m_frame_fft;
    std::vector<std::complex<float>> m_frame_dqpsk;
    std::vector<float> m_impulse_response;
    std::vector<float> m_frequency_response;
    Observable<tcb::span<const viterbi_bit_t>> m_on_frame;
};
