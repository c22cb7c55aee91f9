// dab/algorithms/dab_viterbi_decoder.cpp -- see the header.  Reference: src/dab/algorithms/dab_viterbi_decoder.cpp.
#include "./dab_viterbi_decoder.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "../constants/puncture_codes.h"
#include "../dabgpu_shared_context.h"

DAB_Viterbi_Decoder::DAB_Viterbi_Decoder() : m_ctx(dabgpu_shared_context()) {}
DAB_Viterbi_Decoder::~DAB_Viterbi_Decoder() = default;

// :109-112
void DAB_Viterbi_Decoder::reset(const size_t starting_state) {
    m_start_state = starting_state;
    m_current_decoded_bit = 0;
    m_symbols.clear();
    m_nb_segments = 0;
    m_has_tail = false;
    m_is_bad = false;
    for (int i = 0; i < 4; i++) { m_seg_pi[i] = 0; m_seg_steps[i] = 0; }
}

// which PI_n is this kept-count vector? (the reference passes rows of PI_TABLE or PI_X, puncture_codes.h:42-72)
static int identify_puncture_code(tcb::span<const uint8_t> code) {
    if (code.size() == 6) {
        for (size_t i = 0; i < 6; i++) if (code[i] != 2) return -1;
        return 0;      // PI_X
    }
    if (code.size() != 8) return -1;
    for (int pi = 1; pi <= 24; pi++)
        if (std::memcmp(code.data(), dab_puncture_table().pi[pi - 1], 8) == 0) return pi;
    return -1;
}

// :114-122 + depuncture bookkeeping of :131-181 (no symbols are expanded on the host)
size_t DAB_Viterbi_Decoder::update(tcb::span<const viterbi_bit_t> punctured_symbols, tcb::span<const uint8_t> puncture_code,
                                   const size_t requested_output_symbols) {
    if (requested_output_symbols == 0) return 0;
    const int pi = identify_puncture_code(puncture_code);
    size_t consumed = 0;
    for (size_t g = 0, out = 0; out < requested_output_symbols; out += m_code_rate, g = (g + 1) % puncture_code.size())
        consumed += puncture_code[g];
    const bool is_tail = (pi == 0 && requested_output_symbols == 24);
    const bool is_body = (pi >= 1 && requested_output_symbols % 128 == 0 && m_nb_segments < 4 && !m_has_tail);
    if (consumed > punctured_symbols.size() || (!is_tail && !is_body) || (is_tail && m_has_tail)) {
        m_is_bad = true;          // shape the device kernel does not implement: chainback() reports it
        return 0;
    }
    if (is_tail) {
        m_has_tail = true;
    } else {
        m_seg_pi[m_nb_segments] = (uint32_t)pi;
        m_seg_steps[m_nb_segments] = (uint32_t)(requested_output_symbols / m_code_rate);
        m_nb_segments++;
    }
    m_symbols.insert(m_symbols.end(), punctured_symbols.begin(), punctured_symbols.begin() + (std::ptrdiff_t)consumed);
    m_current_decoded_bit += requested_output_symbols / m_code_rate;
    return consumed;
}

// :124-129
uint64_t DAB_Viterbi_Decoder::chainback(tcb::span<uint8_t> bytes_out, const size_t end_state) {
    if (m_is_bad || !m_has_tail)
        throw std::runtime_error("DAB_Viterbi_Decoder: segment sequence not supported by the device decoder "
                                 "(expected up to 4 PI_n segments of 128*L symbols, then the 24-symbol PI_X tail)");
    uint64_t path_error = 0;
    const int tie = std::getenv("DABGPU_TIE_RULE") ? std::atoi(std::getenv("DABGPU_TIE_RULE")) : 0;
    const int st = dabgpu_viterbi_decode_host_sync(m_ctx, m_symbols.data(), m_symbols.size(), m_seg_pi, m_seg_steps,
                                                   (uint32_t)m_start_state, (uint32_t)end_state, DABGPU_CW_RAW, bytes_out.data(),
                                                   bytes_out.size(), &path_error, tie);
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("DAB_Viterbi_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    return path_error;
}
