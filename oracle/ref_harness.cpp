// ref_harness.cpp -- ORACLE-SIDE TEST INFRASTRUCTURE (not product code).
//
// Thin extern "C" entry points over the parts of the reference that compile from their own
// source files in this image.  This file is compiled TOGETHER WITH reference sources where they
// lie under /root/reference (see oracle/Makefile, target _ref); nothing from the reference is
// copied into this repository and the resulting oracle/_ref/libdab_ref.so is git-ignored.
//
// Buildable here (used to pin the C restatement in oracle/dab_oracle_*.c):
//   src/ofdm/dsp/apply_pll.cpp, src/ofdm/dsp/complex_conj_mul_sum.cpp   (AVX2+FMA and baseline builds)
//   src/ofdm/dab_mapper_ref.cpp, dab_prs_ref.cpp, dab_ofdm_params_ref.cpp (fmt: header-only copy in the image)
//   src/dab/msc/cif_deinterleaver.cpp
//   header-only: additive_scrambler.h, crc.h, puncture_codes.h, subchannel_protection_tables.h,
//                dab_parameters.h, chebyshev_sine.h
// NOT buildable here (no stand-ins are written for them):
//   src/ofdm/ofdm_demodulator.cpp, ofdm_modulator.cpp  -> need <fftw3.h>/libfftw3f (absent)
//   src/dab/algorithms/dab_viterbi_decoder.cpp          -> needs vendor/viterbi_decoder (empty submodule)
//   src/dab/fic/fic_decoder.cpp, msc/msc_decoder.cpp    -> link against the above
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>

#include "utility/span.h"
#include "ofdm/dsp/apply_pll.h"
#include "ofdm/dsp/complex_conj_mul_sum.h"
#include "ofdm/dsp/chebyshev_sine.h"
#include "ofdm/dab_mapper_ref.h"
#include "ofdm/dab_prs_ref.h"
#include "ofdm/dab_ofdm_params_ref.h"
#include "ofdm/ofdm_params.h"
#include "dab/msc/cif_deinterleaver.h"
#include "dab/algorithms/additive_scrambler.h"
#include "dab/algorithms/crc.h"
#include "dab/constants/puncture_codes.h"
#include "dab/constants/subchannel_protection_tables.h"
#include "dab/constants/dab_parameters.h"
#include "dab/database/dab_database_entities.h"

using c32 = std::complex<float>;

// second build of the two DSP translation units without AVX/FMA (symbols renamed on the command line)
void apply_pll_auto_baseline(tcb::span<const c32> x, tcb::span<c32> y, const float freq_norm, const float dt_norm);
c32 complex_conj_mul_sum_auto_baseline(tcb::span<const c32> x0, tcb::span<const c32> x1);

extern "C" {

void ref_apply_pll(const float* x, float* y, size_t n, float f, float dt0) {
    apply_pll_auto({reinterpret_cast<const c32*>(x), n}, {reinterpret_cast<c32*>(y), n}, f, dt0);
}
void ref_apply_pll_baseline(const float* x, float* y, size_t n, float f, float dt0) {
    apply_pll_auto_baseline({reinterpret_cast<const c32*>(x), n}, {reinterpret_cast<c32*>(y), n}, f, dt0);
}
void ref_conj_mul_sum(const float* x0, const float* x1, size_t n, float* out2) {
    const c32 r = complex_conj_mul_sum_auto({reinterpret_cast<const c32*>(x0), n}, {reinterpret_cast<const c32*>(x1), n});
    out2[0] = r.real(); out2[1] = r.imag();
}
void ref_conj_mul_sum_baseline(const float* x0, const float* x1, size_t n, float* out2) {
    const c32 r = complex_conj_mul_sum_auto_baseline({reinterpret_cast<const c32*>(x0), n}, {reinterpret_cast<const c32*>(x1), n});
    out2[0] = r.real(); out2[1] = r.imag();
}
float ref_chebyshev_sine(float x) { return chebyshev_sine(x); }

int ref_get_mapper(int* out, size_t nb_carriers, size_t nb_fft) {
    get_DAB_mapper_ref({out, nb_carriers}, nb_fft);
    return 0;
}
int ref_get_prs(int mode, float* out, size_t nb_fft) {
    try { get_DAB_PRS_reference(mode, {reinterpret_cast<c32*>(out), nb_fft}); } catch (...) { return 1; }
    return 0;
}
int ref_get_ofdm_params(int mode, size_t* out6) {
    try {
        const OFDM_Params p = get_DAB_OFDM_params(mode);
        out6[0] = p.nb_frame_symbols; out6[1] = p.nb_symbol_period; out6[2] = p.nb_null_period;
        out6[3] = p.nb_cyclic_prefix; out6[4] = p.nb_fft; out6[5] = p.nb_data_carriers;
    } catch (...) { return 1; }
    return 0;
}
int ref_get_dab_params(int mode, int* out13) {
    try {
        const DAB_Parameters p = get_dab_parameters(mode);
        const int v[13] = { p.nb_frame_bits, p.nb_symbols, p.nb_fic_symbols, p.nb_msc_symbols, p.nb_fibs, p.nb_cifs,
                            p.nb_fibs_per_cif, p.nb_sym_bits, p.nb_fic_bits, p.nb_msc_bits, p.nb_fib_bits,
                            p.nb_fib_cif_bits, p.nb_cif_bits };
        std::memcpy(out13, v, sizeof(v));
    } catch (...) { return 1; }
    return 0;
}

void* ref_deint_create(int nb_bytes) { return new CIF_Deinterleaver(nb_bytes); }
void ref_deint_destroy(void* h) { delete static_cast<CIF_Deinterleaver*>(h); }
void ref_deint_consume(void* h, const int8_t* bits, size_t n) { static_cast<CIF_Deinterleaver*>(h)->Consume({bits, n}); }
int ref_deint_deinterleave(void* h, int8_t* out, size_t n) { return static_cast<CIF_Deinterleaver*>(h)->Deinterleave({out, n}) ? 1 : 0; }

void ref_scrambler_bytes(uint8_t* out, size_t n) {
    AdditiveScrambler s;
    s.SetSyncword(0xFFFF);      // fic_decoder.cpp:48, msc_decoder.cpp:40
    s.Reset();
    for (size_t i = 0; i < n; i++) out[i] = s.Process();
}
uint16_t ref_crc16(const uint8_t* x, size_t n) {
    static CRC_Calculator<uint16_t>* calc = [] {    // parameters of fic_decoder.cpp:19-31
        auto* c = new CRC_Calculator<uint16_t>(0x1021);
        c->SetInitialValue(0xFFFF);
        c->SetFinalXORValue(0xFFFF);
        return c;
    }();
    return calc->Process({x, n});
}
void ref_puncture_tables(uint8_t* pi_24x8, uint8_t* pi_x6) {
    for (int i = 1; i <= 24; i++) { auto c = GetPunctureCode(i); for (int j = 0; j < 8; j++) pi_24x8[(i-1)*8+j] = c[j]; }
    for (int j = 0; j < 6; j++) pi_x6[j] = PI_X[j];
}
// (PI, Lx) plan exactly as MSC_Decoder::DecodeEEP/DecodeUEP (msc_decoder.cpp:77-94,118-131) derive it
int ref_subchannel_plan(int length, int is_uep, int uep_index, int eep_level, int eep_type_b, int* pi4, int* lx4) {
    Subchannel sc(0);
    sc.length = static_cast<subchannel_size_t>(length);
    sc.is_uep = is_uep != 0;
    sc.uep_prot_index = static_cast<uep_protection_index_t>(uep_index);
    sc.eep_prot_level = static_cast<eep_protection_level_t>(eep_level);
    sc.eep_type = eep_type_b ? EEP_Type::TYPE_B : EEP_Type::TYPE_A;
    if (!sc.is_uep) {
        const auto d = GetEEPDescriptor(sc);
        const int n = sc.length / d.capacity_unit_multiple;
        for (int i = 0; i < EEP_Descriptor::TOTAL_PUNCTURE_CODES; i++) { pi4[i] = d.PIx[i]; lx4[i] = d.Lx[i].GetLx(n); }
        return EEP_Descriptor::TOTAL_PUNCTURE_CODES;
    }
    const auto d = GetUEPDescriptor(sc);
    for (int i = 0; i < UEP_Descriptor::TOTAL_PUNCTURE_CODES; i++) { pi4[i] = d.PIx[i]; lx4[i] = d.Lx[i]; }
    return UEP_Descriptor::TOTAL_PUNCTURE_CODES;
}
int ref_uep_row(int index, int* size_bitrate_level_padding4) {
    const auto& d = UEP_PROTECTION_TABLE[index];
    size_bitrate_level_padding4[0] = d.subchannel_size; size_bitrate_level_padding4[1] = d.bitrate;
    size_bitrate_level_padding4[2] = d.protection_level; size_bitrate_level_padding4[3] = d.total_padding_bits;
    return 0;
}

} // extern "C"
