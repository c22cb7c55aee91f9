// app_helpers/app_viterbi_convert_block.h -- soft-bit <-> packed hard-bit conversion with the reference's names
// (examples/app_helpers/app_viterbi_convert_block.h:12-44), executed on the device through libdabgpu.so.
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>
#include "dab/dabgpu_shared_context.h"
#include "dabgpu.h"
#include "utility/span.h"
#include "viterbi_config.h"

static inline void convert_viterbi_bytes_to_bits(tcb::span<const uint8_t> bytes, tcb::span<viterbi_bit_t> bits) {
    if (bytes.size() * 8 != bits.size()) throw std::runtime_error("convert_viterbi_bytes_to_bits: size mismatch");
    const int st = dabgpu_hard_bytes_to_soft_bits_host_sync(dabgpu_shared_context(), bytes.data(), bytes.size(), bits.data());
    if (st != DABGPU_OK) throw std::runtime_error(std::string("dabgpu_hard_bytes_to_soft_bits_host_sync: ") + dabgpu_last_error());
}

static inline void convert_viterbi_bits_to_bytes(tcb::span<const viterbi_bit_t> bits, tcb::span<uint8_t> bytes) {
    if (bytes.size() * 8 != bits.size()) throw std::runtime_error("convert_viterbi_bits_to_bytes: size mismatch");
    const int st = dabgpu_soft_bits_to_hard_bytes_host_sync(dabgpu_shared_context(), bits.data(), bytes.size(), bytes.data());
    if (st != DABGPU_OK) throw std::runtime_error(std::string("dabgpu_soft_bits_to_hard_bytes_host_sync: ") + dabgpu_last_error());
}
