// viterbi_config.h -- soft-decision bit type shared by the OFDM and DAB halves (reference: src/viterbi_config.h:11-14)
#pragma once
#include <stdint.h>

typedef int8_t viterbi_bit_t;
static constexpr viterbi_bit_t SOFT_DECISION_VITERBI_HIGH = +127;
static constexpr viterbi_bit_t SOFT_DECISION_VITERBI_LOW = -127;
static constexpr viterbi_bit_t SOFT_DECISION_VITERBI_PUNCTURED = 0;
