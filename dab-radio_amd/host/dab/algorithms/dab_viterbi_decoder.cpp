// dab/algorithms/dab_viterbi_decoder.cpp -- see the header.  Reference: src/dab/algorithms/dab_viterbi_decoder.cpp.
#include "./dab_viterbi_decoder.h"

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "dabgpu.h"
#include "../dabgpu_shared_context.h"

DAB_Viterbi_Decoder::DAB_Viterbi_Decoder() : m_ctx(dabgpu_shared_context()) {}
DAB_Viterbi_Decoder::~DAB_Viterbi_Decoder() = default;

// :109-112
void DAB_Viterbi_Decoder::reset(const size_t starting_state) {
    m_start_state = starting_state;
    m_current_decoded_bit = 0;
    m_mother.clear();
}

// :114-122 with depuncture_symbols (:131-181) run here, on the host: ANY puncture vector (kept counts 0..4 per group of four mother
// symbols, applied cyclically), any requested_output_symbols (a multiple of the code rate), any number of calls.  The add-compare-select
// the reference runs inside update() happens in chainback(), on the device, over everything recorded since reset().
size_t DAB_Viterbi_Decoder::update(tcb::span<const viterbi_bit_t> punctured_symbols, tcb::span<const uint8_t> puncture_code,
                                   const size_t requested_output_symbols) {
    if (requested_output_symbols == 0 || puncture_code.empty()) return 0;
    const size_t before = m_mother.size();
    size_t in = 0, code = 0, out = 0;
    while (out < requested_output_symbols) {                          // :154-176
        const size_t keep = puncture_code[code] < m_code_rate ? puncture_code[code] : m_code_rate;
        if (punctured_symbols.size() - in < keep) {                   // :157-160: res comes back as initialised -- nothing decoded, nothing consumed
            m_mother.resize(before);
            return 0;
        }
        for (size_t i = 0; i < keep; i++) m_mother.push_back(punctured_symbols[in++]);
        for (size_t i = keep; i < m_code_rate; i++) m_mother.push_back(0);          // SOFT_DECISION_VITERBI_PUNCTURED
        out += m_code_rate;
        code = (code + 1) % puncture_code.size();
    }
    m_current_decoded_bit += out / m_code_rate;
    return in;
}

// :124-129.  bytes_out.size() * 8 bits are traced back from decision word bytes_out.size() * 8 + 5 downwards, starting in end_state;
// the path error is the accumulated renormalisation + metric[end_state] after every step decoded since reset().
uint64_t DAB_Viterbi_Decoder::chainback(tcb::span<uint8_t> bytes_out, const size_t end_state) {
    if (m_current_decoded_bit == 0)
        throw std::invalid_argument("DAB_Viterbi_Decoder::chainback: nothing was decoded since reset()");
    uint64_t path_error = 0;
    const int tie = dabgpu_core_model_from_env();
    const int st = dabgpu_viterbi_decode_depunctured_host_sync(m_ctx, m_mother.data(), m_current_decoded_bit, (uint32_t)m_start_state,
                                                               (uint32_t)end_state, bytes_out.data(), bytes_out.size(), &path_error, tie);
    if (st == DABGPU_ERR_INVALID_ARG)           // a trace-back that starts beyond the decoded steps (the reference would read stale decision words)
        throw std::invalid_argument(std::string("DAB_Viterbi_Decoder::chainback: ") + dabgpu_last_error());
    if (st != DABGPU_OK)
        throw std::runtime_error(std::string("DAB_Viterbi_Decoder: ") + dabgpu_strerror(st) + " -- " + dabgpu_last_error());
    return path_error;
}
