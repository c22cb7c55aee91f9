// dab/fic/fic_decoder.cpp -- reference: src/dab/fic/fic_decoder.cpp:33-117
#include "./fic_decoder.h"

#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "../dabgpu_frame_batcher.h"
#include "../dabgpu_shared_context.h"

FIC_Decoder::FIC_Decoder(const size_t nb_encoded_bits, const size_t nb_fibs_per_group)
    : m_ctx(dabgpu_shared_context()), m_nb_fibs_per_group(nb_fibs_per_group), m_nb_encoded_bits(nb_encoded_bits) {
    m_decoded_bytes.resize(nb_encoded_bits / 24);        // rate 1/3 after puncturing (:39-41)
    dabgpu_frame_batcher::add_fic();
}

FIC_Decoder::~FIC_Decoder() { dabgpu_frame_batcher::remove_fic(); }

void FIC_Decoder::DecodeFIBGroup(tcb::span<const viterbi_bit_t> encoded_bits, const size_t cif_index) {
    // only the Mode I puncturing (PI_16 x21, PI_15 x3, PI_X) is defined by the standard and by the reference (:61-72)
    if (m_nb_encoded_bits != DABGPU_NB_FIB_GROUP_BITS || m_nb_fibs_per_group != 3 || encoded_bits.size() < m_nb_encoded_bits) return;
    // the group of a frame this process's OFDM_Demod produced was decoded with the rest of that frame (dabgpu_frame_batcher.h)
    if (!dabgpu_frame_batcher::fetch_fib_group(encoded_bits.data(), (int)(cif_index & 3), m_decoded_bytes.data(), &m_last_crc_mask, &m_last_error)) {
        const int st = dabgpu_fic_decode_group_host_sync(m_ctx, encoded_bits.data(), m_decoded_bytes.data(), &m_last_crc_mask,
                                                         &m_last_error, dabgpu_tie_rule_from_env());
        if (st != DABGPU_OK)
            throw std::runtime_error(std::string("FIC_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
        dabgpu_frame_batcher::count_call_by_call(true);
    }
    const size_t fib_bytes = m_decoded_bytes.size() / m_nb_fibs_per_group;
    for (size_t i = 0; i < m_nb_fibs_per_group; i++)                                   // :103-116
        if (m_last_crc_mask & (1u << i))
            m_on_fib.Notify(tcb::span<const uint8_t>(m_decoded_bytes.data() + i * fib_bytes, fib_bytes - 2));
}
