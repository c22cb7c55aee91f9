// dab/msc/cif_deinterleaver.h -- CIF_Deinterleaver with the reference's public interface
// (src/dab/msc/cif_deinterleaver.h:9-22); the 16-CIF ring lives in device memory.
#pragma once
#include "utility/span.h"
#include "viterbi_config.h"

struct dabgpu_msc_stream;

class CIF_Deinterleaver {
public:
    explicit CIF_Deinterleaver(const int nb_bytes);
    ~CIF_Deinterleaver();
    CIF_Deinterleaver(const CIF_Deinterleaver&) = delete;
    CIF_Deinterleaver& operator=(const CIF_Deinterleaver&) = delete;
    void Consume(tcb::span<const viterbi_bit_t> bits_buf);
    bool Deinterleave(tcb::span<viterbi_bit_t> out_bits_buf);

private:
    dabgpu_msc_stream* m_stream;
    const int m_nb_bytes;
};
