// dab/constants/dab_parameters.h -- transmission-frame bit geometry (reference: src/dab/constants/dab_parameters.h:5-90;
// field names are API: basic_radio reads them)
#pragma once
#include <stdexcept>

struct DAB_Parameters {
    int nb_frame_bits;
    int nb_symbols;
    int nb_fic_symbols;
    int nb_msc_symbols;
    int nb_fibs;
    int nb_cifs;
    int nb_fibs_per_cif;
    int nb_sym_bits;
    int nb_fic_bits;
    int nb_msc_bits;
    int nb_fib_bits;
    int nb_fib_cif_bits;
    int nb_cif_bits;
};

static inline DAB_Parameters get_dab_parameters(const int transmission_mode) {
    // DAB_parameters A1.1 / A1.3: {carriers, symbols incl. PRS, FIC symbols, FIBs, CIFs}
    static const int table[4][5] = {{1536, 76, 3, 12, 4}, {384, 76, 3, 3, 1}, {192, 153, 8, 4, 1}, {768, 76, 3, 6, 2}};
    if (transmission_mode < 1 || transmission_mode > 4) throw std::runtime_error("Invalid transmission mode");
    const int* r = table[transmission_mode - 1];
    DAB_Parameters p;
    p.nb_symbols = r[1] - 1;
    p.nb_frame_bits = r[0] * 2 * p.nb_symbols;
    p.nb_fic_symbols = r[2];
    p.nb_msc_symbols = p.nb_symbols - r[2];
    p.nb_fibs = r[3];
    p.nb_cifs = r[4];
    p.nb_fibs_per_cif = r[3] / r[4];
    p.nb_sym_bits = p.nb_frame_bits / p.nb_symbols;
    p.nb_fic_bits = p.nb_sym_bits * p.nb_fic_symbols;
    p.nb_msc_bits = p.nb_sym_bits * p.nb_msc_symbols;
    p.nb_fib_bits = p.nb_fic_bits / p.nb_fibs;
    p.nb_fib_cif_bits = p.nb_fib_bits * p.nb_fibs_per_cif;
    p.nb_cif_bits = p.nb_msc_bits / p.nb_cifs;
    return p;
}
