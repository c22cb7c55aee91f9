// dab/algorithms/dab_viterbi_decoder.h -- DAB_Viterbi_Decoder with the reference's public interface
// (src/dab/algorithms/dab_viterbi_decoder.h:12-33) over the MI355X C ABI.
//
// The reference runs the add-compare-select inside update() and the pointer chase inside chainback(); on the device both live in one
// kernel, so update() only de-punctures (exactly as depuncture_symbols does, :131-181) and records the mother code, and chainback()
// launches the decode of everything recorded since reset().  The values returned by update() (symbols consumed),
// get_current_decoded_bit() and chainback() (path error) are the reference's.  As general as the reference's class: any puncture
// vector, any requested_output_symbols, any number of update() calls, any start and end state, a trace-back shorter than the decoded
// length.  (FIC_Decoder / MSC_Decoder do not go through this class: they call the batch decoders, which de-puncture on the device.)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "utility/span.h"
#include "viterbi_config.h"

struct dabgpu_ctx;

class DAB_Viterbi_Decoder {
public:
    static constexpr size_t m_constraint_length = 7;
    static constexpr size_t m_code_rate = 4;

    DAB_Viterbi_Decoder();
    ~DAB_Viterbi_Decoder();
    DAB_Viterbi_Decoder(const DAB_Viterbi_Decoder&) = delete;
    DAB_Viterbi_Decoder& operator=(const DAB_Viterbi_Decoder&) = delete;

    void set_traceback_length(const size_t traceback_length) { m_traceback_length = traceback_length; }
    size_t get_traceback_length() const { return m_traceback_length; }
    size_t get_current_decoded_bit() const { return m_current_decoded_bit; }
    void reset(const size_t starting_state = 0u);
    size_t update(tcb::span<const viterbi_bit_t> punctured_symbols, tcb::span<const uint8_t> puncture_code,
                  const size_t requested_output_symbols);
    uint64_t chainback(tcb::span<uint8_t> bytes_out, const size_t end_state = 0u);

private:
    dabgpu_ctx* m_ctx;
    size_t m_traceback_length = 0;
    size_t m_current_decoded_bit = 0;
    size_t m_start_state = 0;
    std::vector<viterbi_bit_t> m_mother;      // de-punctured mother code since reset(): 4 soft bits per trellis step, punctured ones 0
};
