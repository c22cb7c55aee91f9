// dab/msc/msc_decoder.cpp -- reference: src/dab/msc/msc_decoder.cpp:26-154
#include "./msc_decoder.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "../dabgpu_frame_batcher.h"
#include "../dabgpu_shared_context.h"

struct MSC_Decoder::BatchState {
    dabgpu_subchannel sc;
    dabgpu_frame_batcher::cif_id last;
    int run = 0, cifs_seen = 0;
    // the last 16 CIFs of the sub-channel as handed in, kept on the HOST until a call has to be decoded call by call: a decoder that only
    // picks its bytes up from its demodulator's batched decodes (the steady state) owns no device context, stream, ring or scratch at all
    std::vector<viterbi_bit_t> ring;
    int next_slot = 0, stored = 0;
};

MSC_Decoder::MSC_Decoder(const Subchannel subchannel)
    : m_subchannel(subchannel), m_stream(nullptr), m_ctx(nullptr), m_batch(new BatchState()) {
    dabgpu_subchannel& sc = m_batch->sc;
    sc.start_address = subchannel.start_address;
    sc.length = subchannel.length;
    sc.is_uep = subchannel.is_uep ? 1 : 0;
    sc.uep_prot_index = subchannel.uep_prot_index;
    sc.eep_prot_level = subchannel.eep_prot_level;
    sc.eep_type = (subchannel.eep_type == EEP_Type::TYPE_B) ? 1 : 0;
    int st = DABGPU_OK;
    try {
        (void)dabgpu_shared_context();                  // no device, no decoder: there is no CPU fallback (throws)
        st = dabgpu_subchannel_validate(&sc);           // what dabgpu_msc_stream_create refuses, without creating anything
    } catch (...) {
        delete m_batch;
        m_batch = nullptr;
        throw;
    }
    if (st != DABGPU_OK) {
        delete m_batch;
        m_batch = nullptr;
        throw std::runtime_error(std::string("MSC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    }
    m_batch->ring.resize((size_t)16 * (size_t)subchannel.length * 64);
    m_decoded_bytes.resize((size_t)subchannel.length * 8);                              // :30
    if (const char* e = std::getenv("DABGPU_MSC_EAGER")) if (std::atoi(e)) EnsureStream();   // development: the device objects at construction, as before round 6
    dabgpu_frame_batcher::add_subchannel(m_batch->sc);
}

MSC_Decoder::~MSC_Decoder() {
    dabgpu_frame_batcher::remove_subchannel(m_batch->sc);
    dabgpu_msc_stream_destroy(m_stream);
    if (m_ctx) dabgpu_destroy(m_ctx);
    delete m_batch;
}

// the first call-by-call decode: a context of this decoder's own (stream + scratch, so that decoders on different threads run side by side:
// src/basic_radio/basic_radio.cpp:51-62) and the device ring, brought up to date with the CIFs filed on the host so far, oldest first
void MSC_Decoder::EnsureStream() {
    if (m_stream) return;
    BatchState& B = *m_batch;
    m_ctx = dabgpu_private_context();
    int st = dabgpu_msc_stream_create(m_ctx, &B.sc, &m_stream);
    const size_t n_bits = (size_t)m_subchannel.length * 64;
    for (int k = 0; k < B.stored && st == DABGPU_OK; k++)
        st = dabgpu_msc_stream_push_cif(m_stream, B.ring.data() + (size_t)((B.next_slot + 16 - B.stored + k) % 16) * n_bits);
    if (st != DABGPU_OK) {
        dabgpu_msc_stream_destroy(m_stream);
        m_stream = nullptr;
        dabgpu_destroy(m_ctx);
        m_ctx = nullptr;
        throw std::runtime_error(std::string("MSC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    }
    std::vector<viterbi_bit_t>().swap(B.ring);          // from here on the stream's own (page-locked) ring files the CIFs
}

tcb::span<uint8_t> MSC_Decoder::DecodeCIF(tcb::span<const viterbi_bit_t> buf) {
    const size_t start_bit = (size_t)m_subchannel.start_address * 64, n_bits = (size_t)m_subchannel.length * 64;
    if (start_bit + n_bits > buf.size()) return {};                                     // :50-54
    BatchState& B = *m_batch;
    if (m_stream) {
        if (dabgpu_msc_stream_push_cif(m_stream, buf.data() + start_bit) != DABGPU_OK) return {};
    } else {                                                                            // cif_deinterleaver.cpp:28-33, on the host
        std::memcpy(B.ring.data() + (size_t)B.next_slot * n_bits, buf.data() + start_bit, n_bits);
        B.next_slot = (B.next_slot + 1) % 16;
        if (B.stored < 16) B.stored++;
    }
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
    if (B.cifs_seen < 16) return {};                                                   // :60-63 (nothing to decode yet: still no device object)
    EnsureStream();
    const int st = dabgpu_msc_stream_decode_sync(m_stream, m_decoded_bytes.data(), &n_out, &m_last_error, dabgpu_tie_rule_from_env());
    if (st == DABGPU_ERR_NOT_READY) return {};                                          // :60-63
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("MSC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    dabgpu_frame_batcher::count_call_by_call(false);
    return tcb::span<uint8_t>(m_decoded_bytes.data(), n_out);
}
