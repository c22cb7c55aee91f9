// dab/fic/fic_decoder.h -- FIC_Decoder with the reference's public interface (src/dab/fic/fic_decoder.h:15-35)
// over the MI355X C ABI: one device launch per FIB group does Viterbi + descramble + CRC16.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "utility/observable.h"
#include "utility/span.h"
#include "viterbi_config.h"

struct dabgpu_ctx;

class FIC_Decoder {
public:
    FIC_Decoder(const size_t nb_encoded_bits, const size_t nb_fibs_per_group);
    ~FIC_Decoder();
    void DecodeFIBGroup(tcb::span<const viterbi_bit_t> encoded_bits, const size_t cif_index);
    auto& OnFIB(void) { return m_on_fib; }
    // last group: path error and per-FIB CRC result (the reference only logs these, fic_decoder.cpp:88,111)
    uint64_t GetLastPathError() const { return m_last_error; }
    uint32_t GetLastCrcMask() const { return m_last_crc_mask; }

private:
    dabgpu_ctx* m_ctx;
    const size_t m_nb_fibs_per_group;
    const size_t m_nb_encoded_bits;
    std::vector<uint8_t> m_decoded_bytes;
    uint64_t m_last_error = 0;
    uint32_t m_last_crc_mask = 0;
    Observable<tcb::span<const uint8_t>> m_on_fib;
};
