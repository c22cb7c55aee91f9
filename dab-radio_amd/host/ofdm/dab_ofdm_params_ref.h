#pragma once
#include "./ofdm_params.h"
// reference: src/ofdm/dab_ofdm_params_ref.cpp:10-58 (throws std::runtime_error on an invalid mode)
OFDM_Params get_DAB_OFDM_params(const int transmission_mode);
