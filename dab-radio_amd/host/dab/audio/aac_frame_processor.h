// dab/audio/aac_frame_processor.h -- AAC_Frame_Processor with the reference's public interface
// (src/dab/audio/aac_frame_processor.h:33-88): DAB+ logical frames in, super-frame header and AAC access units out, with
// the fire-code / Reed-Solomon / access-unit CRC work done on the device (libdabgpu.so, dabgpu_dabplus_*).
#pragma once
#include <stdint.h>
#include <vector>
#include "utility/observable.h"
#include "utility/span.h"

struct dabgpu_dabplus_bank;

enum class MPEG_Surround { NOT_USED, SURROUND_51, SURROUND_71, SURROUND_OTHER, RFA };

struct SuperFrameHeader {
    uint32_t sampling_rate = 0;
    bool is_parametric_stereo = false;
    bool is_spectral_band_replication = false;
    bool is_stereo = false;
    MPEG_Surround mpeg_surround = MPEG_Surround::NOT_USED;
    bool operator==(const SuperFrameHeader& other) const {
        return sampling_rate == other.sampling_rate && is_parametric_stereo == other.is_parametric_stereo &&
               is_spectral_band_replication == other.is_spectral_band_replication && is_stereo == other.is_stereo &&
               mpeg_surround == other.mpeg_surround;
    }
    bool operator!=(const SuperFrameHeader& other) const { return !(*this == other); }
};

class AAC_Frame_Processor {
public:
    AAC_Frame_Processor();
    ~AAC_Frame_Processor();
    AAC_Frame_Processor(const AAC_Frame_Processor&) = delete;
    AAC_Frame_Processor& operator=(const AAC_Frame_Processor&) = delete;
    // An audio super frame consists of 5 DAB logical frames
    void Process(tcb::span<const uint8_t> buf);
    auto& OnFirecodeError(void) { return m_obs_firecode_error; }          // frame_index, crc_got, crc_calculated
    auto& OnRSError(void) { return m_obs_rs_error; }                      // rs_frame_index, rs_total_frames
    auto& OnSuperFrameHeader(void) { return m_obs_superframe_header; }
    auto& OnAccessUnitCRCError(void) { return m_obs_au_crc_error; }       // au_index, total_aus, crc_got, crc_calculated
    auto& OnAccessUnit(void) { return m_obs_access_unit; }                // au_index, total_aus, au_buffer
private:
    dabgpu_dabplus_bank* m_bank = nullptr;
    std::vector<uint8_t> m_super_frame_buf;
    Observable<const int, const uint16_t, const uint16_t> m_obs_firecode_error;
    Observable<const int, const int> m_obs_rs_error;
    Observable<SuperFrameHeader> m_obs_superframe_header;
    Observable<const int, const int, const uint16_t, const uint16_t> m_obs_au_crc_error;
    Observable<const int, const int, tcb::span<uint8_t>> m_obs_access_unit;
};
