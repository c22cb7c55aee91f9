// dab/audio/aac_frame_processor.cpp -- see the header.  Reference cited: src/dab/audio/aac_frame_processor.cpp.
#include "./aac_frame_processor.h"

#include <stdexcept>
#include <string>

#include "dab/dabgpu_shared_context.h"
#include "dabgpu.h"

AAC_Frame_Processor::AAC_Frame_Processor() {
    const int st = dabgpu_dabplus_bank_create(dabgpu_shared_context(), 1, &m_bank);
    if (st != DABGPU_OK) throw std::runtime_error(std::string("AAC_Frame_Processor: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
}

AAC_Frame_Processor::~AAC_Frame_Processor() { dabgpu_dabplus_bank_destroy(m_bank); }

void AAC_Frame_Processor::Process(tcb::span<const uint8_t> buf) {
    const int N = (int)buf.size();
    if (N == 0 || N < 11) return;                                          // :129-137
    m_super_frame_buf.resize((size_t)5 * N);
    int done = 0, wait_failed = 0;
    uint32_t wait_crc = 0;
    dabgpu_superframe_result r;
    const int st = dabgpu_dabplus_process_frame_host_sync(m_bank, buf.data(), (uint32_t)N, &done, &wait_failed, &wait_crc, &r,
                                                          m_super_frame_buf.data());
    if (st != DABGPU_OK) throw std::runtime_error(std::string("AAC_Frame_Processor: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    if (wait_failed) {                                                     // :178-189 in State::WAIT_FRAME_START (no frame collected yet)
        m_obs_firecode_error.Notify(0, (uint16_t)(wait_crc >> 16), (uint16_t)(wait_crc & 0xFFFF));
        return;
    }
    if (!done) return;
    const int n_rs = 5 * N / 120;
    if (r.rs_failed_index >= 0) { m_obs_rs_error.Notify(r.rs_failed_index, n_rs); return; }          // :336-341
    if (!r.firecode_ok) {                                                                             // :206-209, all 5 frames collected
        m_obs_firecode_error.Notify(5, (uint16_t)(r.firecode_rx_calc >> 16), (uint16_t)(r.firecode_rx_calc & 0xFFFF));
        return;
    }
    SuperFrameHeader h;                                                                               // :215-257
    h.sampling_rate = (r.descriptor & 0x40) ? 48000 : 32000;
    h.is_spectral_band_replication = (r.descriptor & 0x20) != 0;
    h.is_stereo = (r.descriptor & 0x10) != 0;
    h.is_parametric_stereo = (r.descriptor & 0x08) != 0;
    switch (r.descriptor & 0x07) {
    case 0: h.mpeg_surround = MPEG_Surround::NOT_USED; break;
    case 1: h.mpeg_surround = MPEG_Surround::SURROUND_51; break;
    case 2: h.mpeg_surround = MPEG_Surround::SURROUND_71; break;
    case 7: h.mpeg_surround = MPEG_Surround::SURROUND_OTHER; break;
    default: h.mpeg_surround = MPEG_Surround::RFA; break;
    }
    m_obs_superframe_header.Notify(h);
    const int walked = (r.au_walk_stopped_at < 0) ? r.num_aus : r.au_walk_stopped_at;
    for (int i = 0; i < walked; i++) {                                                                // :286-317
        const int a = r.au_start[i], nb_data = r.au_start[i + 1] - a - 2;
        if (r.au_crc_ok_mask & (1u << i)) {
            m_obs_access_unit.Notify(i, r.num_aus, tcb::span<uint8_t>(m_super_frame_buf.data() + a, (size_t)nb_data));
        } else {
            const uint16_t rx = (uint16_t)((m_super_frame_buf[(size_t)a + nb_data] << 8) | m_super_frame_buf[(size_t)a + nb_data + 1]);
            m_obs_au_crc_error.Notify(i, r.num_aus, rx, r.au_crc_calc[i]);
        }
    }
}
