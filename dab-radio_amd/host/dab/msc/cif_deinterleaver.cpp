// dab/msc/cif_deinterleaver.cpp -- reference: src/dab/msc/cif_deinterleaver.cpp:13-71
#include "./cif_deinterleaver.h"

#include <stdexcept>

#include "dabgpu.h"
#include "../dabgpu_shared_context.h"

CIF_Deinterleaver::CIF_Deinterleaver(const int nb_bytes) : m_stream(nullptr), m_nb_bytes(nb_bytes) {
    // a ring needs only the size; any protection profile of that size will do for the geometry (EEP 4-A: multiples of 4 CU)
    dabgpu_subchannel sc = {0, nb_bytes / 8, 0, 0, 3, 0};
    if (nb_bytes % 32 != 0 || dabgpu_msc_stream_create(dabgpu_shared_context(), &sc, &m_stream) != DABGPU_OK)
        throw std::runtime_error("CIF_Deinterleaver: sub-channel size must be a whole number of 4-CU blocks");
}
CIF_Deinterleaver::~CIF_Deinterleaver() { dabgpu_msc_stream_destroy(m_stream); }

void CIF_Deinterleaver::Consume(tcb::span<const viterbi_bit_t> bits_buf) {
    if ((int)bits_buf.size() < m_nb_bytes * 8) return;
    dabgpu_msc_stream_push_cif(m_stream, bits_buf.data());
}

bool CIF_Deinterleaver::Deinterleave(tcb::span<viterbi_bit_t> out_bits_buf) {
    if ((int)out_bits_buf.size() < m_nb_bytes * 8) return false;
    return dabgpu_msc_stream_deinterleave_sync(m_stream, out_bits_buf.data()) == DABGPU_OK;
}
