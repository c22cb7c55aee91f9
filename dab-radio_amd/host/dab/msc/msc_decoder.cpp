// dab/msc/msc_decoder.cpp -- reference: src/dab/msc/msc_decoder.cpp:26-154
#include "./msc_decoder.h"

#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "../dabgpu_frame_batcher.h"
#include "../dabgpu_shared_context.h"

struct MSC_Decoder::BatchState {
    dabgpu_subchannel sc;
    dabgpu_frame_batcher::cif_id last;
    int run = 0, cifs_seen = 0;
};

MSC_Decoder::MSC_Decoder(const Subchannel subchannel)
    : m_subchannel(subchannel), m_stream(nullptr), m_ctx(dabgpu_private_context()), m_batch(new BatchState()) {
    dabgpu_subchannel& sc = m_batch->sc;
    sc.start_address = subchannel.start_address;
    sc.length = subchannel.length;
    sc.is_uep = subchannel.is_uep ? 1 : 0;
    sc.uep_prot_index = subchannel.uep_prot_index;
    sc.eep_prot_level = subchannel.eep_prot_level;
    sc.eep_type = (subchannel.eep_type == EEP_Type::TYPE_B) ? 1 : 0;
    const int st = dabgpu_msc_stream_create(m_ctx, &sc, &m_stream);
    if (st != DABGPU_OK) {
        dabgpu_destroy(m_ctx);
        m_ctx = nullptr;
        delete m_batch;
        m_batch = nullptr;
    }
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("MSC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    m_decoded_bytes.resize((size_t)subchannel.length * 8);                              // :30
    dabgpu_frame_batcher::add_subchannel(m_batch->sc);
}

MSC_Decoder::~MSC_Decoder() {
    dabgpu_frame_batcher::remove_subchannel(m_batch->sc);
    dabgpu_msc_stream_destroy(m_stream);
    dabgpu_destroy(m_ctx);
    delete m_batch;
}

tcb::span<uint8_t> MSC_Decoder::DecodeCIF(tcb::span<const viterbi_bit_t> buf) {
    const size_t start_bit = (size_t)m_subchannel.start_address * 64, n_bits = (size_t)m_subchannel.length * 64;
    if (start_bit + n_bits > buf.size()) return {};                                     // :50-54
    if (dabgpu_msc_stream_push_cif(m_stream, buf.data() + start_bit) != DABGPU_OK) return {};
    BatchState& B = *m_batch;
    if (B.cifs_seen < 16) B.cifs_seen++;
    size_t n_out = 0;
    // A CIF of a frame this process's OFDM_Demod produced was decoded with the rest of that frame (dab/dabgpu_frame_batcher.h).  Its
    // result is this decoder's result once the last 16 CIFs it was handed are the batcher's last 16 (the time de-interleaver's span)
    {
        const auto id = dabgpu_frame_batcher::match_cif(buf.data() + start_bit, start_bit, n_bits, B.last);
        const auto next = dabgpu_frame_batcher::successor(B.last);
        B.run = !id.valid() ? 0 : ((B.last.valid() && next.src == id.src && next.gen == id.gen && next.cif == id.cif) ? B.run + 1 : 1);
        if (B.run > 16) B.run = 16;
        B.last = id;
        if (id.valid() && B.run >= 16 && B.cifs_seen >= 16 &&
            dabgpu_frame_batcher::fetch_cif(id, B.sc, m_decoded_bytes.data(), m_decoded_bytes.size(), &n_out, &m_last_error))
            return tcb::span<uint8_t>(m_decoded_bytes.data(), n_out);
    }
    const int st = dabgpu_msc_stream_decode_sync(m_stream, m_decoded_bytes.data(), &n_out, &m_last_error, dabgpu_tie_rule_from_env());
    if (st == DABGPU_ERR_NOT_READY) return {};                                          // :60-63
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("MSC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    dabgpu_frame_batcher::count_call_by_call(false);
    return tcb::span<uint8_t>(m_decoded_bytes.data(), n_out);
}
