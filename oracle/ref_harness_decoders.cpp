// ref_harness_decoders.cpp -- ORACLE-SIDE TEST INFRASTRUCTURE (not product code).
//
// The reference's own channel-decoder CONTROL code, executed: src/dab/fic/fic_decoder.cpp (:53-117 DecodeFIBGroup: the three update() calls with
// PI_16 / PI_15 / PI_X, chain-back, descrambler, CRC16, OnFIB), src/dab/msc/msc_decoder.cpp (:46-154 DecodeCIF / DecodeEEP / DecodeUEP: slicing, the
// time de-interleaver, the (PI, L) segments of the protection tables, the padding bits, chain-back, descrambler) and src/dab/msc/cif_deinterleaver.cpp
// are compiled from the reference tree IN PLACE (oracle/Makefile, target _ref/libdab_ref_decoders.so; nothing is copied, the library is git-ignored).
//
// What they call -- class DAB_Viterbi_Decoder, declared in the reference's src/dab/algorithms/dab_viterbi_decoder.h:12-45 -- cannot be built from the
// reference here: dab_viterbi_decoder.cpp includes vendor/viterbi_decoder (williamyang98/ViterbiDecoderCpp), an empty submodule.  THIS FILE DEFINES
// THAT CLASS OVER THE ORACLE'S RESTATED CORE (oracle/dab_oracle_decode.c: dab_viterbi_*), so:
//   * the vectors made with this library (tests/golden/decoder_vectors.npz) pin rows a20-a22 of SURVEY 8 -- which bits each decoder feeds the core,
//     in which order, with which puncturing vectors and lengths, what it does with the decoded bytes -- to EXECUTED reference code;
//   * they do NOT pin the add-compare-select core or the de-puncturing loop of dab_viterbi_decoder.cpp:131-181: those stay restated
//     ("parity unpinned").  The fixtures say so in their `label` field.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "utility/span.h"
#include "viterbi_config.h"
#include "dab/algorithms/dab_viterbi_decoder.h"
#include "dab/database/dab_database_entities.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/msc_decoder.h"

extern "C" {
#include "dab_oracle.h"
}

static int g_core_model = 0;           // which upstream core the restated one models (dab_oracle.h: 0 scalar, 1 SIMD), read at construction

class DAB_Viterbi_Decoder_Internal {
public:
    dab_viterbi* v = nullptr;
    size_t traceback_length = 0;
    int core_model = 0;
    ~DAB_Viterbi_Decoder_Internal() { dab_viterbi_destroy(v); }
};

DAB_Viterbi_Decoder::DAB_Viterbi_Decoder() : m_accumulated_error(0) {
    m_decoder = std::make_unique<DAB_Viterbi_Decoder_Internal>();
    m_decoder->core_model = g_core_model;
}
DAB_Viterbi_Decoder::~DAB_Viterbi_Decoder() = default;
void DAB_Viterbi_Decoder::set_traceback_length(const size_t traceback_length) {
    dab_viterbi_destroy(m_decoder->v);
    m_decoder->v = dab_viterbi_create(traceback_length, m_decoder->core_model);
    m_decoder->traceback_length = traceback_length;
}
size_t DAB_Viterbi_Decoder::get_traceback_length() const { return m_decoder->traceback_length; }
size_t DAB_Viterbi_Decoder::get_current_decoded_bit() const { return dab_viterbi_current_decoded_bit(m_decoder->v); }
void DAB_Viterbi_Decoder::reset(const size_t starting_state) { dab_viterbi_reset(m_decoder->v, starting_state); }
size_t DAB_Viterbi_Decoder::update(tcb::span<const viterbi_bit_t> punctured_symbols, tcb::span<const uint8_t> puncture_code, const size_t requested_output_symbols) {
    return dab_viterbi_update(m_decoder->v, punctured_symbols.data(), punctured_symbols.size(), puncture_code.data(), puncture_code.size(), requested_output_symbols);
}
uint64_t DAB_Viterbi_Decoder::chainback(tcb::span<uint8_t> bytes_out, const size_t end_state) {
    return dab_viterbi_chainback(m_decoder->v, bytes_out.data(), bytes_out.size(), end_state);
}

namespace {
struct FicHandle {
    std::unique_ptr<FIC_Decoder> dec;
    std::vector<uint8_t> fibs;          // the FIBs OnFIB delivered during the current call, back to back
    size_t fib_bytes = 0;
};
}  // namespace

extern "C" {

void ref_dec_set_core_model(int model) { g_core_model = model ? 1 : 0; }

void* ref_fic_create(size_t nb_encoded_bits, size_t nb_fibs_per_group) {
    auto* h = new FicHandle();
    h->dec = std::make_unique<FIC_Decoder>(nb_encoded_bits, nb_fibs_per_group);
    h->dec->OnFIB().Attach([h](tcb::span<const uint8_t> fib) { h->fib_bytes = fib.size(); h->fibs.insert(h->fibs.end(), fib.begin(), fib.end()); });
    return h;
}
void ref_fic_destroy(void* p) { delete static_cast<FicHandle*>(p); }
// FIC_Decoder::DecodeFIBGroup(bits, cif_index); returns the number of FIBs OnFIB delivered, their bytes (30 each) in out
int ref_fic_decode_group(void* p, const int8_t* bits, size_t n_bits, size_t cif_index, uint8_t* out, size_t cap) {
    auto* h = static_cast<FicHandle*>(p);
    h->fibs.clear();
    h->dec->DecodeFIBGroup({bits, n_bits}, cif_index);
    if (h->fibs.size() > cap) return -1;
    std::memcpy(out, h->fibs.data(), h->fibs.size());
    return h->fib_bytes ? (int)(h->fibs.size() / h->fib_bytes) : 0;
}

void* ref_msc_create(int id, int start_address, int length, int is_uep, int uep_index, int eep_level, int eep_type_b) {
    Subchannel sc(static_cast<subchannel_id_t>(id));
    sc.start_address = static_cast<subchannel_addr_t>(start_address);
    sc.length = static_cast<subchannel_size_t>(length);
    sc.is_uep = is_uep != 0;
    sc.uep_prot_index = static_cast<uep_protection_index_t>(uep_index);
    sc.eep_prot_level = static_cast<eep_protection_level_t>(eep_level);
    sc.eep_type = eep_type_b ? EEP_Type::TYPE_B : EEP_Type::TYPE_A;
    sc.is_complete = true;
    return new MSC_Decoder(sc);
}
void ref_msc_destroy(void* p) { delete static_cast<MSC_Decoder*>(p); }
// MSC_Decoder::DecodeCIF(cif); returns the size of the span it returned (0 while the de-interleaver fills / on its error paths), bytes in out
long ref_msc_decode_cif(void* p, const int8_t* cif, size_t n_bits, uint8_t* out, size_t cap) {
    auto got = static_cast<MSC_Decoder*>(p)->DecodeCIF({cif, n_bits});
    if (got.size() > cap) return -1;
    std::memcpy(out, got.data(), got.size());
    return (long)got.size();
}

}  // extern "C"
