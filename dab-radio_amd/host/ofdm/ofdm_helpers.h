// ofdm/ofdm_helpers.h -- Create_OFDM_Demodulator as in the reference (src/ofdm/ofdm_helpers.h:12-20)
#pragma once
#include <complex>
#include <memory>
#include <vector>
#include "./dab_mapper_ref.h"
#include "./dab_ofdm_params_ref.h"
#include "./dab_prs_ref.h"
#include "./ofdm_demodulator.h"
#include "./ofdm_params.h"

static inline std::unique_ptr<OFDM_Demod> Create_OFDM_Demodulator(const int transmission_mode, const int total_threads = 0) {
    const OFDM_Params params = get_DAB_OFDM_params(transmission_mode);
    std::vector<std::complex<float>> prs(params.nb_fft);
    std::vector<int> mapper(params.nb_data_carriers);
    get_DAB_PRS_reference(transmission_mode, prs);
    get_DAB_mapper_ref(mapper, params.nb_fft);
    return std::make_unique<OFDM_Demod>(params, prs, mapper, total_threads);
}
