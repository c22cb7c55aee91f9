#pragma once
#include <complex>
#include "utility/span.h"
// reference: src/ofdm/dab_prs_ref.cpp:140-195 (throws std::runtime_error on invalid mode / too small buffer)
void get_DAB_PRS_reference(const int transmission_mode, tcb::span<std::complex<float>> buf);
