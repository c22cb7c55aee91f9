// ofdm/ofdm_demodulator.h -- OFDM_Demod with the reference's public interface (src/ofdm/ofdm_demodulator.h:24-167),
// re-implemented over the MI355X C ABI (include/dabgpu.h).
//
// What runs where:
//   host  : the 5-state framing machine (Process, :235-275), sample buffering, the L1 power statistics used only to
//           find the first NULL symbol (:291-347, :922-950) -- control flow over a few scalars per 100 samples
//   device: coarse frequency sync + fine time sync, PLL + cyclic-prefix phase + 77 x FFT + DQPSK + de-interleave + soft bits +
//           fine-frequency loop, and -- when FIC_Decoder / MSC_Decoder objects of this process listen (dab/dabgpu_frame_batcher.h) -- the
//           frame's FIC and sub-channel decode: all ENQUEUED per frame through the receiver pipeline (dabgpu_receiver_*, include/dabgpu.h),
//           the frequency state resident on the device.
// Threads (the reference's reader / coordinator / pipeline threads, ofdm_demodulator.cpp:550-639, as two):
//   reader   = the caller of Process(): frames the stream into page-locked staging buffers and submits; it waits for the device in two
//              places only -- for the synchroniser's record once it has buffered on to where the earliest possible frame would end (or
//              when the next Process() begins), and for a free slot when `depth` frames are already in flight (:565 WaitEnd)
//   delivery = one internal thread, the coordinator's role (:621-638): waits for frame k's results, publishes the getters' values, calls
//              the On_OFDM_Frame observers -- serialised, in frame order, while the reader buffers and submits frame k + 1
// Ordering contract: per frame, coarse sync -> fine time sync -> demodulation -> fine-frequency update; frame k + 1's synchroniser sees
// frame k's update (device stream order).  The getters return the state as of the frame delivered last (and the resets since): inside
// an observer they are that frame's values, whatever the reader has submitted meanwhile.  Synchronize() (not in the reference) returns
// when every frame handed to the device has been delivered -- call it before reading the counters at the end of a stream; the
// destructor does.  DABGPU_MIRROR_DEPTH = frames in flight (1..6, default 3; 1 = the reference's double buffer).
// nb_desired_threads is accepted for source compatibility and ignored (the device kernel replaces the thread pool).
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <complex>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>
#include "utility/observable.h"
#include "utility/span.h"
#include "viterbi_config.h"
#include "./ofdm_params.h"

struct dabgpu_receiver;

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
    void Synchronize();
#define DABGPU_MIRROR_HAS_SYNCHRONIZE 1

    OFDM_Params GetOFDMParams() const { return m_params; }
    State GetState() const { return m_state; }
    auto& GetConfig() { return m_cfg; }
    const auto& GetConfig() const { return m_cfg; }
    // The getters may be called from any thread (a GUI polls them, src/.../render_ofdm_demod.cpp): every value is published by the delivery
    // thread as ONE atomic word, so none can be read torn; inside an On_OFDM_Frame observer they are that frame's values.  After the last
    // Process() call they are final once Synchronize() (or the destructor) has returned: frames are delivered from the delivery thread.
    float GetSignalAverage() const { return m_signal_l1_average; }
    float GetFineFrequencyOffset() const { return m_freq_fine.load(std::memory_order_relaxed); }
    float GetCoarseFrequencyOffset() const { return m_freq_coarse.load(std::memory_order_relaxed); }
    float GetNetFrequencyOffset() const { return m_freq_fine.load(std::memory_order_relaxed) + m_freq_coarse.load(std::memory_order_relaxed); }
    int GetFineTimeOffset() const { return m_fine_time_offset.load(std::memory_order_relaxed); }
    int GetTotalFramesRead() const { return m_total_frames_read.load(std::memory_order_relaxed); }
    int GetTotalFramesDesync() const { return m_total_frames_desync.load(std::memory_order_relaxed); }
    // GUI views (FFT of all 77 symbols, per-carrier DQPSK vectors): the reference fills them for every frame (ofdm_demodulator.cpp:701-709,734); here
    // they cost two extra device-to-host copies per frame (2.2 MB), so they are fetched from the first call of either getter on -- a viewer linked
    // unchanged sees zeros for the frames before its first poll and the frame's data afterwards.  EnableDebugBuffers(true) before Process() gives the
    // reference's behaviour from the first frame; EnableDebugBuffers(false) stops the copies until a getter is called again.
    tcb::span<const std::complex<float>> GetFrameFFT() const { m_fetch_debug.store(true, std::memory_order_relaxed); return m_frame_fft; }
    tcb::span<const std::complex<float>> GetFrameDataVec() const { m_fetch_debug.store(true, std::memory_order_relaxed); return m_frame_dqpsk; }
    // the newest delivered frame's soft bits: a view into one of the receiver's 8 result slots -- valid until 7 more frames have been submitted (the
    // reference's view is overwritten by the NEXT frame, ofdm_demodulator.cpp:627-638: callers copy, app_ofdm_blocks.h:32-35); the length never changes
    tcb::span<const viterbi_bit_t> GetFrameDataBits() const { return {m_bits_ptr.load(std::memory_order_acquire), m_bits_len}; }
    tcb::span<const float> GetImpulseResponse() const { return m_impulse_response; }
    tcb::span<const float> GetCoarseFrequencyResponse() const { return m_frequency_response; }
    tcb::span<const std::complex<float>> GetCorrelationTimeBuffer() const { return m_corr; }
    auto& On_OFDM_Frame() { return m_on_frame; }
    void EnableDebugBuffers(bool enable) { m_fetch_debug.store(enable, std::memory_order_relaxed); }

private:
    size_t FindNullPowerDip(tcb::span<const std::complex<float>> buf);
    size_t ReadNullPRS(tcb::span<const std::complex<float>> buf);
    size_t ReadSymbols(tcb::span<const std::complex<float>> buf);
    void Run(tcb::span<const std::complex<float>> buf);
    void SubmitSync();
    void CollectPendingSync();
    void ReplayAfterFailedSync(size_t buffered);
    bool ResolveSync();
    void SubmitFrame();
    void ResetReader();
    float L1Average(const std::complex<float>* block, size_t n) const;
    void UpdateSignalAverage(tcb::span<const std::complex<float>> block);
    void DeliveryThread();
    void RethrowDeliveryError();

    OFDM_Demod_Config m_cfg;
    State m_state = FINDING_NULL_POWER_DIP;
    const OFDM_Params m_params;
    int m_mode = 0;                          // transmission mode 1..4 matching m_params
    dabgpu_receiver* m_rx = nullptr;
    bool m_banked = false;                   // a member of the device's receiver bank (csrc/receiver_bank.hip)
    bool m_counted = false;                  // this object is in the count of live mode I receivers (the receiver bank's AUTO rule)
    std::atomic<int>* m_live = nullptr;
    // ---- reader side ----
    bool m_null_start_found = false;
    bool m_null_end_found = false;
    float m_signal_l1_average = 0.0f;
    mutable std::atomic<bool> m_fetch_debug{false};    // set by GetFrameFFT() / GetFrameDataVec() / EnableDebugBuffers
    // null search ring (nb_null_period), NULL+PRS correlation window
    std::vector<std::complex<float>> m_ring;
    size_t m_ring_index = 0, m_ring_length = 0;
    std::vector<std::complex<float>> m_corr;
    size_t m_corr_length = 0;
    // the staging buffer under assembly (page-locked, the receiver's): NULL | PRS | ... ; samples buffered, where the frame ends
    std::complex<float>* m_stage = nullptr;
    size_t m_stage_length = 0, m_stage_capacity = 0, m_frame_end = 0;
    bool m_sync_pending = false;             // the synchroniser's record has not been collected yet
    // while the record is outstanding: where each Process() call's samples begin in the staging buffer and the signal average in force for them
    struct Segment { size_t start; float average; };
    std::vector<Segment> m_spec_segments;
    bool m_replaying = false;                // ReplayAfterFailedSync is running the serial machine over what was buffered
    int m_reader_time_offset = 0;
    uint64_t m_subs_version = ~0ull;         // batcher subscription the receiver was last given
    int m_depth = 3;
    // ---- published by the delivery thread, in submission order ----
    std::atomic<int> m_total_frames_read{0};
    std::atomic<int> m_total_frames_desync{0};
    std::atomic<float> m_freq_coarse{0.0f};
    std::atomic<float> m_freq_fine{0.0f};
    std::atomic<int> m_fine_time_offset{0};
    std::atomic<const viterbi_bit_t*> m_bits_ptr{nullptr};
    size_t m_bits_len = 0;                                   // (set once in the constructor: the frame length of the mode)
    std::vector<viterbi_bit_t> m_frame_bits;                 // what GetFrameDataBits() shows before the first frame
    std::vector<std::complex<float>> m_frame_fft;
    std::vector<std::complex<float>> m_frame_dqpsk;
    std::vector<float> m_impulse_response;
    std::vector<float> m_frequency_response;
    Observable<tcb::span<const viterbi_bit_t>> m_on_frame;
    // ---- reader -> delivery ----
    struct Item { enum Kind { SYNC, RESET, FRAME } kind; float coarse, fine; int offset; uint64_t gen; bool views; bool batch; };
    std::mutex m_mu;
    std::condition_variable m_cv_items, m_cv_done;
    std::deque<Item> m_items;
    int m_frames_in_flight = 0;
    bool m_busy = false, m_stop = false;
    std::exception_ptr m_error;
    std::thread m_thread;
    // ---- reader -> decode thread: the generations whose demodulation is enqueued; it waits (host) for each and enqueues the frame's decode ----
    std::deque<uint64_t> m_to_decode;
    uint64_t m_decodes_submitted = 0;        // generations 0 .. m_decodes_submitted - 1 have their decode enqueued
    uint64_t m_frames_submitted = 0;         // generations handed to the device by the reader
    std::condition_variable m_cv_decode, m_cv_decoded;
    std::thread m_decode_thread;
    void DecodeThread();
    // DABGPU_MIRROR_PROFILE=1: where the two threads spend their time, microseconds, printed by the destructor
    bool m_profile = false;
    double m_t_sync_wait = 0, m_t_slot_wait = 0, m_t_submit = 0, m_t_frame_wait = 0, m_t_observers = 0, m_t_batcher = 0;
};
