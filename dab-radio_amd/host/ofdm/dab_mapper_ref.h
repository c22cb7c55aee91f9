#pragma once
#include <stddef.h>
#include "utility/span.h"
// reference: src/ofdm/dab_mapper_ref.cpp:10-51
void get_DAB_mapper_ref(tcb::span<int> carrier_map, const size_t nb_fft);
