// ofdm/dab_refs.cpp -- Mode parameters, PRS spectrum and carrier map, host side.
// Mode I tables come from the C ABI (the same tables the device context uses); modes II-IV are geometry only
// (the kernels are Mode I; see DESIGN.md "out of scope").
#include <stdexcept>
#include <string>
#include <vector>

#include "dabgpu.h"
#include "./dab_mapper_ref.h"
#include "./dab_ofdm_params_ref.h"
#include "./dab_prs_ref.h"

OFDM_Params get_DAB_OFDM_params(const int transmission_mode) {
    // DAB_parameters A1.1 (reference: src/ofdm/dab_ofdm_params_ref.cpp:13-52): {symbols, symbol period, null period, fft, carriers}
    static const size_t table[4][5] = {
        {76, 2552, 2656, 2048, 1536}, {76, 638, 664, 512, 384}, {153, 319, 345, 256, 192}, {76, 1276, 1328, 1024, 768}};
    if (transmission_mode < 1 || transmission_mode > 4)
        throw std::runtime_error("Invalid transmission mode " + std::to_string(transmission_mode));
    const size_t* r = table[transmission_mode - 1];
    OFDM_Params p;
    p.nb_frame_symbols = r[0];
    p.nb_symbol_period = r[1];
    p.nb_null_period = r[2];
    p.nb_fft = r[3];
    p.nb_cyclic_prefix = r[1] - r[3];
    p.nb_data_carriers = r[4];
    return p;
}

void get_DAB_PRS_reference(const int transmission_mode, tcb::span<std::complex<float>> buf) {
    if (transmission_mode < 1 || transmission_mode > 4)
        throw std::runtime_error("Invalid transmission mode " + std::to_string(transmission_mode));
    if (transmission_mode != 1)
        throw std::runtime_error("dabgpu: only the Mode I phase reference symbol is built in");
    if (buf.size() < DABGPU_NB_DATA_CARRIERS + 1)
        throw std::runtime_error("FFT buffer not large enough to fit phase reference symbol");
    std::vector<float> tmp(2 * DABGPU_NB_FFT);
    if (buf.size() != DABGPU_NB_FFT || dabgpu_get_prs_fft_ref(1, tmp.data()) != DABGPU_OK)
        throw std::runtime_error("dabgpu: Mode I PRS needs a 2048-bin buffer");
    for (size_t i = 0; i < buf.size(); i++) buf[i] = std::complex<float>(tmp[2 * i], tmp[2 * i + 1]);
}

void get_DAB_mapper_ref(tcb::span<int> carrier_map, const size_t nb_fft) {
    if (nb_fft != DABGPU_NB_FFT || carrier_map.size() != DABGPU_NB_DATA_CARRIERS ||
        dabgpu_get_carrier_mapper(1, carrier_map.data()) != DABGPU_OK)
        throw std::runtime_error("dabgpu: only the Mode I carrier map (2048 bins, 1536 carriers) is built in");
}
