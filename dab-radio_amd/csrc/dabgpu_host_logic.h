// dabgpu_host_logic.h -- the part of libdabgpu.so that never touches the device: constant tables, protection-profile plans, codeword
// validation, the mapping cost model, run-length rules, capture-format and wav-header parsing, error text.  Plain C++ (no HIP headers):
// compiled into the library by the same Makefile, and on its own with -fsanitize=address,undefined by tests/test_host_sanitizers.py,
// which fuzzes it (tests/cpp/host_logic_fuzz.cpp).
#pragma once
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>

#include "dabgpu.h"

#if defined(__HIPCC__)
#define DABGPU_HD __host__ __device__
#else
#define DABGPU_HD
#endif

void dabgpu_set_error(const char* fmt, ...);          // thread-local detail of the last failure (dabgpu_last_error)

struct dabgpu_msc_plan {            // device-side sub-channel plan (one per sub-channel of the multiplex)
    uint32_t start_address;         // CUs
    uint32_t n_steps;               // trellis steps incl. tail
    uint32_t seg_pi[4];
    uint32_t seg_steps[4];
    uint32_t out_offset;            // byte offset of this sub-channel inside one CIF's output record
    uint32_t n_out_bytes;
    uint32_t lane_mapped;           // this call decodes the sub-channel with the lane-per-codeword kernel: viterbi_kernel skips it
};
struct dabgpu_vit_tables {          // constant tables of the Viterbi kernels, built on the host at context creation
    uint16_t pi_tab[25 * 8];        // [PI][group]: kept count | running prefix << 8 (puncture_codes.h:42-67)
    unsigned char prbs[512];        // energy-dispersal bytes, period 511 (additive_scrambler.h:16-35)
};

// geometry of the four transmission modes (src/ofdm/dab_ofdm_params_ref.cpp:11-60)
namespace dabgpu {
struct ModeGeom { int n_sym, period, null_period, n_fft, n_cp, n_carriers, frame_samples, sym_bits, frame_bits; };

DABGPU_HD inline bool mode_geometry(int mode, ModeGeom& g) {
    switch (mode) {
    case 1: g.n_sym = 76; g.period = 2552; g.null_period = 2656; g.n_fft = 2048; g.n_carriers = 1536; break;
    case 2: g.n_sym = 76; g.period = 638; g.null_period = 664; g.n_fft = 512; g.n_carriers = 384; break;
    case 3: g.n_sym = 153; g.period = 319; g.null_period = 345; g.n_fft = 256; g.n_carriers = 192; break;
    case 4: g.n_sym = 76; g.period = 1276; g.null_period = 1328; g.n_fft = 1024; g.n_carriers = 768; break;
    default: return false;
    }
    g.n_cp = g.period - g.n_fft;
    g.frame_samples = g.n_sym * g.period + g.null_period;
    g.sym_bits = 2 * g.n_carriers;
    g.frame_bits = (g.n_sym - 1) * g.sym_bits;
    return true;
}

}  // namespace dabgpu

// ---- sizes of the batch decoders' scratch areas ----
// decisions are stored two steps per row pair; the chain-back reads whole 24-step chunks
DABGPU_HD static inline uint32_t dabgpu_vit_alloc_steps(uint32_t n_steps) { return (n_steps + 6u + 63u) & ~63u; }
#define DABGPU_VIT_SCHED_PREFETCH 6u          // entries past the last step the trellis loops may load (viterbi_lanes.hip, viterbi_octet.hip)
static_assert(((0u + 6u + 63u) & ~63u) >= 0u + DABGPU_VIT_SCHED_PREFETCH && ((58u + 6u + 63u) & ~63u) >= 58u + DABGPU_VIT_SCHED_PREFETCH,
              "dabgpu_vit_alloc_steps must leave room for the schedule prefetch");
// soft bits a codeword consumes (dab_viterbi_decoder.cpp:131-181): 8 + PI per 8 steps, 12 for the tail
DABGPU_HD static inline uint32_t dabgpu_vit_in_bytes(const uint32_t* seg_pi, const uint32_t* seg_steps) {
    uint32_t n = 12;
    for (int k = 0; k < 4; k++) n += (seg_steps[k] >> 3) * (8u + seg_pi[k]);
    return n;
}
// symbol rows: 4 kept soft bits per row, + the row the last step's two-row window reaches into + one the prefetch may touch
DABGPU_HD static inline uint32_t dabgpu_vit_in_rows(uint32_t n_in) { return (n_in + 3u) / 4u + 2u; }

// symbols_per_block = 0 of the mode I demodulator
int dabgpu_host_small_batch_spb(size_t n_frames);
int dabgpu_host_spb_bucket(size_t n_frames);
int dabgpu_host_spb_variant(int src, int bits_layout, bool tail);
// argument check of one dabgpu_codeword (index i only for the message)
int dabgpu_host_validate_codeword(const dabgpu_codeword& d, size_t i);
// DABGPU_VIT_MAP_AUTO: cost model of the three decoder mappings (n_simd = SIMDs of the device); forced_mapping != AUTO is returned as is
int dabgpu_host_choose_mapping(int forced_mapping, double n_simd, size_t n_cw, size_t n_groups, double sum_cw_steps, double sum_group_steps,
                               double max_steps, bool staged_gather);
// the same for the MSC of n_ens ensembles sharing a multiplex (steps[j] = trellis steps of sub-channel j); model_us[3] (may be null) = the
// modelled WAVE / LANE / OCTET times
int dabgpu_host_choose_msc_mapping(int forced_mapping, double n_simd, size_t n_ens, const uint32_t* steps, int n_sub, double* model_us);
// the device-side plans of a multiplex's sub-channels (msc_decoder.cpp:77-154): DABGPU_OK, or DABGPU_ERR_INVALID_ARG for an invalid
// profile / a sub-channel outside the 864 capacity units / more than 64 sub-channels
int dabgpu_host_build_msc_plans(const dabgpu_subchannel* subs, int n_sub, std::vector<dabgpu_msc_plan>& plans, uint32_t* cif_out_bytes,
                                uint32_t* max_steps, uint32_t* max_out_bytes);
void dabgpu_host_fill_vit_tables(dabgpu_vit_tables* T);
