// ofdm/dab_refs.cpp -- Mode parameters, PRS spectrum and carrier map, host side.
// All tables come from the C ABI (the same tables the device context uses), transmission modes I-IV.
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
    int g[9];
    if (dabgpu_get_ofdm_params(transmission_mode, g) != DABGPU_OK)
        throw std::runtime_error("Invalid transmission mode " + std::to_string(transmission_mode));
    if ((int)buf.size() < g[5] + 1) throw std::runtime_error("FFT buffer not large enough to fit phase reference symbol");
    if ((int)buf.size() != g[3]) throw std::runtime_error("dabgpu: the PRS of a transmission mode needs a buffer of its FFT size");
    std::vector<float> tmp(2 * (size_t)g[3]);
    if (dabgpu_get_prs_fft_ref(transmission_mode, tmp.data()) != DABGPU_OK) throw std::runtime_error(dabgpu_last_error());
    for (size_t i = 0; i < buf.size(); i++) buf[i] = std::complex<float>(tmp[2 * i], tmp[2 * i + 1]);
}

void get_DAB_mapper_ref(tcb::span<int> carrier_map, const size_t nb_fft) {
    for (int mode = 1; mode <= 4; mode++) {
        int g[9];
        if (dabgpu_get_ofdm_params(mode, g) == DABGPU_OK && (size_t)g[3] == nb_fft && (size_t)g[5] == carrier_map.size()) {
            if (dabgpu_get_carrier_mapper(mode, carrier_map.data()) != DABGPU_OK) throw std::runtime_error(dabgpu_last_error());
            return;
        }
    }
    throw std::runtime_error("dabgpu: carrier maps exist for the DAB transmission modes I-IV (fft size / carrier count pairs) only");
}
