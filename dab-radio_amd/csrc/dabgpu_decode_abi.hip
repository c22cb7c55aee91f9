// dabgpu_decode_abi.hip -- channel-decode entry points of the C ABI (include/dabgpu.h): protection-profile
// tables, codeword descriptors, scratch sizing and launches.  Host side only; the arithmetic is in viterbi.hip.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"

static int device_waves(dabgpu_ctx* c) {
    if (c->n_cu <= 0) {
        hipDeviceProp_t p;
        c->n_cu = (hipGetDeviceProperties(&p, c->device) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    return c->n_cu * 32;          // 8 waves per SIMD: the decoder is a serial recurrence per wave, throughput = waves in flight
}

// kept-count vectors PI_1..PI_24 (ETSI EN 300 401 table 13: PI_n keeps 8+n of every 32 mother bits, the e-th extra bit in
// 4-bit group bitrev3(e mod 8)) as count | prefix << 8, and the energy-dispersal PRBS x^9+x^5+1 seeded with all ones
static int ensure_vit_tables(dabgpu_ctx* c) {
    if (c->d_vit_tables) return DABGPU_OK;
    dabgpu_vit_tables T;
    dabgpu_host_fill_vit_tables(&T);
    int st = dabgpu_check_hip(hipMalloc((void**)&c->d_vit_tables, sizeof(T)), "hipMalloc(vit tables)");
    if (st) return st;
    return dabgpu_check_hip(hipMemcpy(c->d_vit_tables, &T, sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(vit tables)");
}

// slot_off: the FIC entry points keep their device scratch in slots of their own (+ FIC_SLOTS), so that one context can decode the
// FIC and the MSC of a batch concurrently on two streams
static const int FIC_SLOTS = 20;
static int run_viterbi(dabgpu_ctx* c, const dabgpu_cw_desc* d_descs, size_t n, uint32_t max_steps, uint32_t max_out_bytes,
                       int tie_rule, dabgpu_codeword_result* d_results, hipStream_t s, int slot_off = 0, size_t n_first = 0,
                       dabgpu_codeword_result* d_results_rest = nullptr) {
    int st0 = ensure_vit_tables(c);
    if (st0) return st0;
    const int n_waves = (int)std::min<size_t>(n, (size_t)device_waves(c));
    const size_t words = ((size_t)max_steps + 63) & ~(size_t)63;
    uint64_t* d_scratch = nullptr;
    int st = dabgpu_scratch(c, 11 + slot_off, (size_t)n_waves * words * sizeof(uint64_t), (void**)&d_scratch, s);
    if (st) return st;
    return dabgpu_check_hip(dabgpu_launch_viterbi(d_descs, (int)n, d_scratch, words, n_waves, (int)max_out_bytes, d_results,
                                                  tie_rule ? 1 : 0, c->d_vit_tables, s, (int)n_first, d_results_rest), "viterbi_kernel launch");
}

extern "C" int dabgpu_viterbi_set_mapping(dabgpu_ctx* c, int mapping) {
    if (!c || mapping < DABGPU_VIT_MAP_AUTO || mapping > DABGPU_VIT_MAP_OCTET) { dabgpu_set_error("viterbi_set_mapping: bad argument"); return DABGPU_ERR_INVALID_ARG; }
    c->vit_mapping = mapping;
    return DABGPU_OK;
}

// AUTO: a cost model of the three mappings on this part (profiles/r01/ab_notes.md, profiles/r03/ab_notes.md; microseconds).
//   WAVE   one wavefront per codeword keeps every SIMD busy: t = sum over codewords of (0.0189 ns x steps + 0.038 us)
//   LANE   a group of 64 codewords is one wavefront that needs 0.5 us per trellis step however many of its lanes are used, and
//          a SIMD works through its groups at that same rate: t = 0.5 us x max(longest schedule, rounds x mean steps) with
//          rounds = ceil(groups / SIMDs), + the gather pass (3.3e-3 / 8.5e-3 us per codeword-kilostep, staged / byte-wise)
//   OCTET  a group is 8 wavefronts of ~57 instructions per step; two of them share a SIMD at 4 cycles per instruction (a lone one
//          issues at half rate, so one costs what two cost): t = 0.095 us x max(2 x longest schedule, rounds8 x mean steps) with
//          rounds8 = ceil(8 groups / SIMDs), + the same gather pass
// n_cw codewords in n_groups groups; sums and maximum of their trellis steps.  Returns DABGPU_VIT_MAP_WAVE / _LANE / _OCTET
static int choose_mapping(dabgpu_ctx* c, size_t n_cw, size_t n_groups, double sum_cw_steps, double sum_group_steps, double max_steps,
                          bool staged_gather) {
    return dabgpu_host_choose_mapping(c->vit_mapping, (double)device_waves(c) / 8.0, n_cw, n_groups, sum_cw_steps, sum_group_steps, max_steps, staged_gather);
}

// lane-per-codeword decoder over prepared groups; the symbol / decision scratch of a launch is bounded (<= 768 bytes per decision
// row: 6 GiB by default, DABGPU_VIT_SCRATCH_MB in the environment overrides), larger batches run as several launches over
// consecutive groups
static size_t lanes_max_rows() {
    size_t mb = 6144;
    if (const char* e = getenv("DABGPU_VIT_SCRATCH_MB")) { const long v = atol(e); if (v > 0) mb = (size_t)v; }
    return std::max<size_t>(mb * 1024 * 1024 / 768, 1);
}

// sym_rows / dec_rows: rows of 64 dwords (kept soft bits, 4 per lane and row) and of 128 dwords (decisions, one row per step)
static int run_viterbi_lanes(dabgpu_ctx* c, const dabgpu_cw_desc* d_descs, const dabgpu_vit_group* d_groups, size_t n_groups,
                             size_t sym_rows, size_t dec_rows, uint32_t max_in_rows, int tie_rule, int ring4, const uint2* d_sched,
                             int octet, dabgpu_codeword_result* d_results, hipStream_t s, int slot_off = 0, uint32_t groups_per_sub = 0) {
    int st;
    uint32_t *d_sym = nullptr, *d_dec = nullptr;
    if ((st = dabgpu_scratch(c, 18 + slot_off, sym_rows * 64 * sizeof(uint32_t), (void**)&d_sym, s))) return st;
    if ((st = dabgpu_scratch(c, 19 + slot_off, dec_rows * 128 * sizeof(uint32_t), (void**)&d_dec, s))) return st;
    return dabgpu_check_hip(dabgpu_launch_viterbi_lanes(d_groups, n_groups, max_in_rows, d_descs, d_sym, d_dec, d_results,
                                                        tie_rule ? 1 : 0, ring4, c->d_vit_tables, d_sched, octet, device_waves(c) / 32, groups_per_sub, s),
                            "vit_lanes_kernel launch");
}

// one puncturing schedule for a whole batch (FIC, uniform codeword batches): groups of 64 consecutive codewords, in bounded slices
static int run_lanes_uniform(dabgpu_ctx* c, const dabgpu_cw_desc* d_descs, size_t n, uint32_t n_steps, const uint32_t* seg_pi,
                             const uint32_t* seg_steps, int tie_rule, int ring4, int octet, dabgpu_codeword_result* d_results, hipStream_t s,
                             int slot_off) {
    int st = ensure_vit_tables(c);
    if (st) return st;
    const uint32_t dec_rows = dabgpu_vit_alloc_steps(n_steps), in_rows = dabgpu_vit_in_rows(dabgpu_vit_in_bytes(seg_pi, seg_steps));
    uint2* d_sched = nullptr;
    if ((st = dabgpu_scratch(c, 25 + slot_off, (size_t)dec_rows * sizeof(uint2), (void**)&d_sched, s))) return st;
    if ((st = dabgpu_check_hip(dabgpu_launch_vit_sched_uniform(d_sched, dec_rows, seg_pi, seg_steps, c->d_vit_tables, s), "vit_sched launch"))) return st;
    const size_t slice_groups = std::max<size_t>(1, lanes_max_rows() / dec_rows);
    for (size_t cw0 = 0; cw0 < n; cw0 += slice_groups * 64) {
        const size_t n_cw = std::min(n - cw0, slice_groups * 64), n_groups = (n_cw + 63) / 64;
        dabgpu_vit_group* d_groups = nullptr;
        if ((st = dabgpu_scratch(c, 17 + slot_off, n_groups * sizeof(dabgpu_vit_group), (void**)&d_groups, s))) return st;
        if ((st = dabgpu_check_hip(dabgpu_launch_vit_groups_uniform(d_groups, n_cw, n_steps, seg_pi, seg_steps, s), "vit_groups launch"))) return st;
        if ((st = run_viterbi_lanes(c, d_descs + cw0, d_groups, n_groups, n_groups * in_rows, n_groups * dec_rows, in_rows, tie_rule, ring4,
                                    d_sched, octet, d_results + cw0, s, slot_off))) return st;
    }
    return DABGPU_OK;
}

extern "C" int dabgpu_viterbi_decode_batch(dabgpu_ctx* c, const dabgpu_codeword* h_cw, size_t n, int tie_rule,
                                           dabgpu_codeword_result* d_results, void* stream) {
    if (!c || (!h_cw && n) || (!d_results && n)) { dabgpu_set_error("viterbi_decode_batch: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n == 0) return DABGPU_OK;
    uint32_t max_steps = 0;
    for (size_t i = 0; i < n; i++) {
        int st = dabgpu_host_validate_codeword(h_cw[i], i);
        if (st) return st;
        max_steps = std::max(max_steps, h_cw[i].n_steps);
    }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    dabgpu_cw_desc* d_descs = nullptr;
    int st = dabgpu_scratch(c, 10, n * sizeof(dabgpu_cw_desc), (void**)&d_descs, s);
    if (st) return st;
    if ((st = dabgpu_stage_h2d(c, d_descs, h_cw, n * sizeof(dabgpu_cw_desc), s))) return st;
    // (h_cw is consumed when this returns, the caller may reuse it: small tables go through the pinned staging ring, large ones through
    // the runtime's pageable path -- or, when the caller's array is page-locked, a copy that is waited for; dabgpu_stage_h2d)
    bool uniform = true;                                  // one puncturing schedule for the whole batch?
    for (size_t i = 0; i < n && uniform; i++) uniform = !(h_cw[i].flags & DABGPU_CW_DEPUNCTURED);      // (mother-code sources: wave mapping only)
    for (size_t i = 1; i < n && uniform; i++)
        uniform = h_cw[i].n_steps == h_cw[0].n_steps && !memcmp(h_cw[i].seg_pi, h_cw[0].seg_pi, sizeof(h_cw[0].seg_pi)) &&
                  !memcmp(h_cw[i].seg_steps, h_cw[0].seg_steps, sizeof(h_cw[0].seg_steps));
    for (size_t i = 0; i < n && uniform; i++)             // the lane mapping keeps ring offsets in 32 bits
        uniform = h_cw[i].n_slots == 0 || (uint64_t)(h_cw[i].n_slots / h_cw[i].cifs_per_frame + 1) * h_cw[i].frame_stride +
                                          (uint64_t)h_cw[i].cifs_per_frame * h_cw[i].cif_stride < ((uint64_t)1 << 32);
    const int map = uniform ? choose_mapping(c, n, (n + 63) / 64, (double)n * max_steps, (double)((n + 63) / 64) * max_steps, (double)max_steps, false)
                            : DABGPU_VIT_MAP_WAVE;
    if (map != DABGPU_VIT_MAP_WAVE)
        return run_lanes_uniform(c, d_descs, n, max_steps, h_cw[0].seg_pi, h_cw[0].seg_steps, tie_rule, 0, map == DABGPU_VIT_MAP_OCTET, d_results, s, 0);
    return run_viterbi(c, d_descs, n, max_steps, max_steps > 6 ? (max_steps - 6) / 8 : 0, tie_rule, d_results, s);
}

static int fic_decode_any(dabgpu_ctx* c, const int8_t* d_bits, size_t n_frames, size_t frame_stride, const int32_t* d_slots,
                          uint8_t* d_fib_bytes, dabgpu_codeword_result* d_results, int tie_rule, void* stream) {
    if (!c || !d_bits || !d_fib_bytes || !d_results) { dabgpu_set_error("fic_decode_frames: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames == 0) return DABGPU_OK;
    if (frame_stride < DABGPU_NB_FIC_BITS) { dabgpu_set_error("fic_decode_frames: frame_stride %zu < 9216", frame_stride); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = n_frames * 4;
    dabgpu_cw_desc* d_descs = nullptr;
    int st = dabgpu_scratch(c, 10 + FIC_SLOTS, n * sizeof(dabgpu_cw_desc), (void**)&d_descs, s);
    if (st) return st;
    st = dabgpu_check_hip(dabgpu_launch_fic_build(d_descs, d_bits, n_frames, frame_stride, d_fib_bytes, d_slots, s), "fic_build_descs launch");
    if (st) return st;
    // FIB groups are contiguous runs of 2304 soft bits; with 16-byte aligned frames the staged gather applies (mode 3)
    const int fic_direct = (((uintptr_t)d_bits % 16 == 0) && (frame_stride % 16 == 0)) ? 3 : 0;
    const int map = choose_mapping(c, n, (n + 63) / 64, (double)n * 774.0, (double)((n + 63) / 64) * 774.0, 774.0, fic_direct != 0);
    if (map != DABGPU_VIT_MAP_WAVE) {
        // one schedule for every FIB group
        const uint32_t seg_pi[4] = {16, 15, 0, 0}, seg_steps[4] = {32 * 21, 32 * 3, 0, 0};
        return run_lanes_uniform(c, d_descs, n, 774, seg_pi, seg_steps, tie_rule, fic_direct, map == DABGPU_VIT_MAP_OCTET, d_results, s, FIC_SLOTS);
    }
    return run_viterbi(c, d_descs, n, 774, 96, tie_rule, d_results, s, FIC_SLOTS);
}

extern "C" int dabgpu_fic_decode_frames(dabgpu_ctx* c, const int8_t* d_bits, size_t n_frames, size_t frame_stride,
                                        uint8_t* d_fib_bytes, dabgpu_codeword_result* d_results, int tie_rule, void* stream) {
    return fic_decode_any(c, d_bits, n_frames, frame_stride, nullptr, d_fib_bytes, d_results, tie_rule, stream);
}

extern "C" int dabgpu_fic_decode_ring(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, const int32_t* d_newest_slot,
                                      uint8_t* d_fib_bytes, dabgpu_codeword_result* d_results, int tie_rule, void* stream) {
    if (!d_newest_slot) { dabgpu_set_error("fic_decode_ring: null slot array"); return DABGPU_ERR_INVALID_ARG; }
    return fic_decode_any(c, d_hist, n_ens, ens_stride, d_newest_slot, d_fib_bytes, d_results, tie_rule, stream);
}

// fic: also decode the FIC of the newest frame of every ensemble (dabgpu_decode_frames_layout).  When every sub-channel runs in a batch
// mapping and the call is one slice, the FIB groups join the MSC launch as further groups of 64 codewords with their own schedule: the
// MSC's groups rarely fill the last round of wavefront slots (4096 ensembles x 18 sub-channels = 4608 groups on 5120 slots), so the FIC
// then costs its 20 us gather and nothing else.  Otherwise it is decoded first, by the FIC entry point's own path.
struct fic_request { uint8_t* d_fib_bytes; dabgpu_codeword_result* d_results; };
static int msc_decode_any(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                          int newest_frame_slot, const int32_t* d_slots, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_out,
                          size_t out_ens_stride, dabgpu_codeword_result* d_results, int tie_rule, void* stream,
                          int bits_layout = DABGPU_BITS_NATURAL, const fic_request* fic = nullptr) {
    if (!c || !d_hist || !h_sub || !d_out || !d_results) { dabgpu_set_error("msc_decode_frames: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (bits_layout != DABGPU_BITS_NATURAL && bits_layout != DABGPU_BITS_MSC_CLASSED) {
        dabgpu_set_error("msc_decode_frames: unknown bits_layout %d", bits_layout); return DABGPU_ERR_INVALID_ARG;
    }
    const int classed = bits_layout == DABGPU_BITS_MSC_CLASSED;
    if (n_ens == 0 || n_sub == 0) return DABGPU_OK;
    if (hist_frames < 5 || newest_frame_slot < 0 || newest_frame_slot >= hist_frames || n_sub < 0) {
        dabgpu_set_error("msc_decode_frames: history_frames must be >= 5 (16 CIFs of delay + the 4 new ones) and 0 <= newest < history_frames");
        return DABGPU_ERR_INVALID_ARG;
    }
    std::vector<dabgpu_msc_plan> plans;
    uint32_t off = 0, max_steps = 0, max_out = 0;
    {
        const int pst = dabgpu_host_build_msc_plans(h_sub, n_sub, plans, &off, &max_steps, &max_out);
        if (pst) return pst;
    }
    if (out_ens_stride < (size_t)4 * off) { dabgpu_set_error("msc_decode_frames: out_ensemble_stride %zu < 4 x %u", out_ens_stride, off); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = n_ens * 4 * (size_t)n_sub;
    const size_t n_fic = fic ? n_ens * 4 : 0;                   // FIB groups of the newest frames (appended to the MSC's descriptors)
    const int8_t* fic_bits = d_slots ? d_hist : d_hist + (size_t)newest_frame_slot * DABGPU_NB_FRAME_BITS;
    dabgpu_cw_desc* d_descs = nullptr;
    dabgpu_msc_plan* d_plans = nullptr;
    int st = dabgpu_scratch(c, 10, (n + n_fic) * sizeof(dabgpu_cw_desc), (void**)&d_descs, s);
    if (st) return st;
    if ((st = dabgpu_scratch(c, 12, plans.size() * sizeof(dabgpu_msc_plan), (void**)&d_plans, s))) return st;
    // (the plans are staged and the descriptors built once the mapping of every sub-channel is known, below)
    // Which sub-channels go to the lane-per-codeword kernel?  The k longest can be left to viterbi_kernel (one wavefront per
    // codeword) and the rest given to vit_lanes_kernel in the same call; AUTO only compares the two pure choices k = 0 and
    // k = n_sub with the cost model of use_lane_mapping() -- a partial viterbi_kernel launch is a single lockstep round of
    // wavefronts and measured 2x its share of a full one, so hybrids did not pay (DABGPU_VIT_HYBRID_K forces one, for the tests).
    // (the lane mapping keeps ring offsets in 32 bits: one ensemble's ring must stay below 4 GiB)
    std::vector<int> order((size_t)n_sub);
    for (int k = 0; k < n_sub; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return plans[(size_t)a].n_steps > plans[(size_t)b].n_steps; });
    const double n_simd = (double)device_waves(c) / 8.0;
    int k_wave = n_sub;                                        // number of (longest) sub-channels left to viterbi_kernel
    int octet = 0;                                             // the others: eight lanes per codeword instead of one
    if ((uint64_t)hist_frames * 230400u < ((uint64_t)1 << 32) && c->vit_mapping != DABGPU_VIT_MAP_WAVE) {
        std::vector<uint32_t> steps((size_t)n_sub);
        for (int j = 0; j < n_sub; j++) steps[(size_t)j] = plans[(size_t)j].n_steps;
        const int m = dabgpu_host_choose_msc_mapping(c->vit_mapping, n_simd, n_ens, steps.data(), n_sub, nullptr);
        if (m != DABGPU_VIT_MAP_WAVE) { k_wave = 0; octet = m == DABGPU_VIT_MAP_OCTET; }
        if (c->vit_mapping == DABGPU_VIT_MAP_AUTO)
            if (const char* e = getenv("DABGPU_VIT_HYBRID_K")) { const int v = atoi(e); if (v >= 0 && v <= n_sub) k_wave = v; }   // tests
    }
    const int n_lane = n_sub - k_wave;
    // the FIC inside the MSC launch?  (every sub-channel in a batch mapping, one slice -- checked again below where the slices are known)
    bool fic_inside = fic != nullptr && k_wave == 0 && n_lane > 0;
    if (fic_inside) {
        size_t rows = 0;
        for (int j = 0; j < n_sub; j++) rows += dabgpu_vit_alloc_steps(plans[(size_t)j].n_steps);
        rows += dabgpu_vit_alloc_steps(774);           // 16 ensembles = 64 FIB groups = one more group of codewords in the launch's scratch
        fic_inside = n_ens <= std::max<size_t>(1, lanes_max_rows() / rows) * 16;
    }
    // Everything in the one-wavefront-per-code-word mapping (a handful of ensembles: one receiver behind the classes is 4 + 72 code words):
    // the FIB groups ride in the sub-channels' launch as code words n .. n + n_fic - 1 with their own result array -- the two launches used
    // to run one after the other on the stream, 109 + 213 us for one ensemble of 18 sub-channels, for work that is independent
    const bool fic_with_wave = fic != nullptr && !fic_inside && k_wave == n_sub &&
                               dabgpu_host_choose_mapping(c->vit_mapping, n_simd, n_fic, (n_fic + 63) / 64, (double)n_fic * 774.0, (double)((n_fic + 63) / 64) * 774.0, 774.0, true) == DABGPU_VIT_MAP_WAVE;
    if (fic && !fic_inside && !fic_with_wave && (st = fic_decode_any(c, fic_bits, n_ens, ens_stride, d_slots, fic->d_fib_bytes, fic->d_results, tie_rule, stream))) return st;
    // flag the lane-mapped sub-channels in the plans the descriptor builder reads, then stage the plans and build the descriptors
    for (int j = k_wave; j < n_sub; j++) plans[(size_t)order[(size_t)j]].lane_mapped = 1;
    if ((st = dabgpu_stage_h2d_cached(c, 0, d_plans, plans.data(), plans.size() * sizeof(dabgpu_msc_plan), s))) return st;
    // (the FIB groups' descriptors behind the sub-channels' in the same launch when they join the sub-channels' trellis launch)
    const bool fic_descs_here = fic_inside || fic_with_wave;
    if (fic_descs_here)
        st = dabgpu_check_hip(dabgpu_launch_msc_fic_build(d_descs, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, d_plans, n_sub, d_out, out_ens_stride,
                                                          (int)off, d_slots, classed, fic_bits, fic->d_fib_bytes, s), "msc_fic_build_descs launch");
    else
        st = dabgpu_check_hip(dabgpu_launch_msc_build(d_descs, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, d_plans, n_sub,
                                                      d_out, out_ens_stride, (int)off, d_slots, classed, s), "msc_build_descs launch");
    if (st) return st;
    if (n_lane > 0) {
        // group (li, gq) = lane-mapped sub-channel li of ensemble-CIFs 64 gq .. 64 gq + 63; ensembles are sliced so that a launch
        // stays inside the scratch bound
        if ((st = ensure_vit_tables(c))) return st;
        size_t dec_rows_per_gq = 0, sym_rows_per_gq = 0;
        uint32_t lane_max_steps = 0, lane_max_in_rows = 0;
        std::vector<uint64_t> lane_subs((size_t)3 * n_lane);             // sub-channel, decision rows before it, symbol rows before it
        for (int j = 0; j < n_lane; j++) {
            const int sidx = order[(size_t)(k_wave + j)];
            const dabgpu_msc_plan& P = plans[(size_t)sidx];
            const uint32_t in_rows = dabgpu_vit_in_rows(dabgpu_vit_in_bytes(P.seg_pi, P.seg_steps));
            lane_subs[(size_t)3 * j] = (uint64_t)sidx;
            lane_subs[(size_t)3 * j + 1] = dec_rows_per_gq;
            lane_subs[(size_t)3 * j + 2] = sym_rows_per_gq;
            dec_rows_per_gq += dabgpu_vit_alloc_steps(P.n_steps);
            sym_rows_per_gq += in_rows;
            lane_max_steps = std::max(lane_max_steps, P.n_steps);
            lane_max_in_rows = std::max(lane_max_in_rows, in_rows);
        }
        const size_t max_gq = std::max<size_t>(1, lanes_max_rows() / dec_rows_per_gq);
        const size_t ens_per_slice = max_gq * 16;                       // 16 ensembles x 4 CIFs = one group per sub-channel
        uint64_t* d_lane_subs = nullptr;
        if ((st = dabgpu_scratch(c, 24, lane_subs.size() * sizeof(uint64_t), (void**)&d_lane_subs, s))) return st;
        if ((st = dabgpu_stage_h2d_cached(c, 1, d_lane_subs, lane_subs.data(), lane_subs.size() * sizeof(uint64_t), s))) return st;
        // the schedule table of every lane-mapped sub-channel, once per call
        const uint32_t sched_stride = dabgpu_vit_alloc_steps(lane_max_steps);
        const uint32_t fic_pi[4] = {16, 15, 0, 0}, fic_steps[4] = {32 * 21, 32 * 3, 0, 0};
        const uint32_t fic_dec_rows = dabgpu_vit_alloc_steps(774), fic_in_rows = dabgpu_vit_in_rows(dabgpu_vit_in_bytes(fic_pi, fic_steps));
        uint2* d_sched = nullptr;
        if ((st = dabgpu_scratch(c, 25, ((size_t)n_lane * sched_stride + (fic_inside ? fic_dec_rows : 0)) * sizeof(uint2), (void**)&d_sched, s))) return st;
        if ((st = dabgpu_check_hip(dabgpu_launch_vit_sched_msc(d_sched, sched_stride, d_plans, d_lane_subs, n_lane, c->d_vit_tables, s), "vit_sched launch"))) return st;
        for (size_t e0 = 0; e0 < n_ens; e0 += ens_per_slice) {
            const size_t ne = std::min(n_ens - e0, ens_per_slice);
            const uint32_t gps = (uint32_t)((ne * 4 + 63) / 64);
            const size_t n_groups = (size_t)n_lane * gps;
            const size_t n_fic_groups = fic_inside ? (n_fic + 63) / 64 : 0;
            dabgpu_vit_group* d_groups = nullptr;
            if ((st = dabgpu_scratch(c, 17, (n_groups + n_fic_groups) * sizeof(dabgpu_vit_group), (void**)&d_groups, s))) return st;
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_groups_msc(d_groups, d_plans, d_lane_subs, n_lane, n_sub, ne, gps, sched_stride, s), "vit_groups launch"))) return st;
            const size_t cw0 = e0 * 4 * (size_t)n_sub;
            // the staged gathers read the ring rows in aligned 16-byte chunks (natural order) / aligned 64-byte lines (class order)
            // (class order: whole 64-byte memory lines are loaded -- with the history and every ensemble 64-byte aligned no line reaches
            // past the end of a row, 230400 = 3600 x 64)
            const int ring4 = classed ? ((((uintptr_t)d_hist % 64 == 0) && (ens_stride % 64 == 0)) ? 2 : 0)
                                      : ((((uintptr_t)d_hist % 16 == 0) && (ens_stride % 16 == 0)) ? 1 : 0);
            if (!fic_inside) {
                if ((st = run_viterbi_lanes(c, d_descs + cw0, d_groups, n_groups, sym_rows_per_gq * gps, dec_rows_per_gq * gps, lane_max_in_rows,
                                            tie_rule, ring4, d_sched, octet, d_results + cw0, s, 0, gps))) return st;
                continue;
            }
            // one slice (e0 = 0): the FIB groups of the newest frames behind the MSC's groups -- descriptors n .., schedule, symbol and
            // decision areas behind the MSC's, results into the caller's FIC array
            const size_t sym_rows = sym_rows_per_gq * gps, dec_rows = dec_rows_per_gq * gps;
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_sched_uniform(d_sched + (size_t)n_lane * sched_stride, fic_dec_rows, fic_pi, fic_steps, c->d_vit_tables, s),
                                       "vit_sched launch"))) return st;
            dabgpu_vit_group_base base;
            base.first = (uint32_t)n;
            base.sched_off = (uint64_t)n_lane * sched_stride;
            base.sym_off = (uint64_t)sym_rows * 64;
            base.dec_off = (uint64_t)dec_rows * 128;
            base.res_delta = (int64_t)(reinterpret_cast<const char*>(fic->d_results) - reinterpret_cast<const char*>(d_results + n));
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_groups_uniform_at(d_groups + n_groups, n_fic, 774, fic_pi, fic_steps, base, s), "vit_groups launch"))) return st;
            uint32_t *d_sym = nullptr, *d_dec = nullptr;
            if ((st = dabgpu_scratch(c, 18, (sym_rows + n_fic_groups * fic_in_rows) * 64 * sizeof(uint32_t), (void**)&d_sym, s))) return st;
            if ((st = dabgpu_scratch(c, 19, (dec_rows + n_fic_groups * fic_dec_rows) * 128 * sizeof(uint32_t), (void**)&d_dec, s))) return st;
            // FIB groups are contiguous runs of 2304 soft bits; with 16-byte aligned frames the staged gather applies
            const int fic_kind = (((uintptr_t)fic_bits % 16 == 0) && (ens_stride % 16 == 0)) ? 3 : 0;
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_prep(ring4, d_groups, n_groups, lane_max_in_rows, d_descs, d_sym, gps, s), "vit_prep launch"))) return st;
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_prep(fic_kind, d_groups + n_groups, n_fic_groups, fic_in_rows, d_descs, d_sym, 0, s), "vit_prep launch"))) return st;
            if ((st = dabgpu_check_hip(dabgpu_launch_vit_trellis(d_groups, n_groups + n_fic_groups, d_descs, d_sym, d_dec, d_results, tie_rule ? 1 : 0,
                                                                 c->d_vit_tables, d_sched, octet, device_waves(c) / 32, s), "vit_lanes_kernel launch"))) return st;
        }
        if (k_wave == 0) return DABGPU_OK;
    }
    if (fic_with_wave)
        return run_viterbi(c, d_descs, n + n_fic, std::max(max_steps, 774u), std::max(max_out, 96u), tie_rule, d_results, s, 0, n, fic->d_results);
    return run_viterbi(c, d_descs, n, max_steps, max_out, tie_rule, d_results, s);
}

// which mapping dabgpu_msc_decode_frames* / dabgpu_decode_frames_layout take for this multiplex and batch right now (the context's setting, or
// the cost model's choice under DABGPU_VIT_MAP_AUTO), and what the model expects of each
extern "C" int dabgpu_multiplex_mapping(dabgpu_ctx* c, size_t n_ens, const dabgpu_subchannel* h_sub, int n_sub, int* mapping, double* model_us3) {
    if (!c || !h_sub || n_sub <= 0 || !mapping) { dabgpu_set_error("multiplex_mapping: null argument"); return DABGPU_ERR_INVALID_ARG; }
    std::vector<dabgpu_msc_plan> plans;
    const int pst = dabgpu_host_build_msc_plans(h_sub, n_sub, plans, nullptr, nullptr, nullptr);
    if (pst) return pst;
    std::vector<uint32_t> steps((size_t)n_sub);
    for (int j = 0; j < n_sub; j++) steps[(size_t)j] = plans[(size_t)j].n_steps;
    *mapping = dabgpu_host_choose_msc_mapping(c->vit_mapping, (double)device_waves(c) / 8.0, n_ens, steps.data(), n_sub, model_us3);
    return DABGPU_OK;
}

extern "C" int dabgpu_msc_decode_frames(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                        int newest_frame_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_out,
                                        size_t out_ens_stride, dabgpu_codeword_result* d_results, int tie_rule, void* stream) {
    return msc_decode_any(c, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, nullptr, h_sub, n_sub, d_out, out_ens_stride,
                          d_results, tie_rule, stream);
}

extern "C" int dabgpu_msc_decode_frames_layout(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                               int newest_frame_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_out,
                                               size_t out_ens_stride, dabgpu_codeword_result* d_results, int tie_rule, int bits_layout,
                                               void* stream) {
    return msc_decode_any(c, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, nullptr, h_sub, n_sub, d_out, out_ens_stride,
                          d_results, tie_rule, stream, bits_layout);
}

extern "C" int dabgpu_msc_decode_ring_layout(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                             const int32_t* d_newest_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_out,
                                             size_t out_ens_stride, dabgpu_codeword_result* d_results, int tie_rule, int bits_layout,
                                             void* stream) {
    if (!d_newest_slot) { dabgpu_set_error("msc_decode_ring: null slot array"); return DABGPU_ERR_INVALID_ARG; }
    return msc_decode_any(c, d_hist, n_ens, ens_stride, hist_frames, 0, d_newest_slot, h_sub, n_sub, d_out, out_ens_stride, d_results,
                          tie_rule, stream, bits_layout);
}

// FIC + MSC of one transmission frame of every ensemble in one call (what BasicRadio::Process fans out to its FIC runner and MSC
// runners, src/basic_radio/basic_radio.cpp:41-65)
extern "C" int dabgpu_decode_frames_layout(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                           int newest_frame_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_fib_bytes,
                                           dabgpu_codeword_result* d_fic_results, uint8_t* d_msc_out, size_t out_ens_stride,
                                           dabgpu_codeword_result* d_msc_results, int tie_rule, int bits_layout, void* stream) {
    if (!c || !d_hist || !d_fib_bytes || !d_fic_results) { dabgpu_set_error("decode_frames: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (ens_stride < DABGPU_NB_FIC_BITS) { dabgpu_set_error("decode_frames: ensemble_stride %zu < 9216", ens_stride); return DABGPU_ERR_INVALID_ARG; }
    if (n_ens == 0) return DABGPU_OK;
    if (hist_frames < 1 || newest_frame_slot < 0 || newest_frame_slot >= hist_frames) {
        dabgpu_set_error("decode_frames: 0 <= newest_frame_slot < history_frames"); return DABGPU_ERR_INVALID_ARG;
    }
    if (n_sub == 0)       // nothing but the FIC
        return dabgpu_fic_decode_frames(c, d_hist + (size_t)newest_frame_slot * DABGPU_NB_FRAME_BITS, n_ens, ens_stride, d_fib_bytes, d_fic_results, tie_rule, stream);
    const fic_request fic = {d_fib_bytes, d_fic_results};
    return msc_decode_any(c, d_hist, n_ens, ens_stride, hist_frames, newest_frame_slot, nullptr, h_sub, n_sub, d_msc_out, out_ens_stride,
                          d_msc_results, tie_rule, stream, bits_layout, &fic);
}

extern "C" int dabgpu_decode_ring_layout(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                         const int32_t* d_newest_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_fib_bytes,
                                         dabgpu_codeword_result* d_fic_results, uint8_t* d_msc_out, size_t out_ens_stride,
                                         dabgpu_codeword_result* d_msc_results, int tie_rule, int bits_layout, void* stream) {
    if (!c || !d_hist || !d_fib_bytes || !d_fic_results || !d_newest_slot) { dabgpu_set_error("decode_ring: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (ens_stride < DABGPU_NB_FIC_BITS) { dabgpu_set_error("decode_ring: ensemble_stride %zu < 9216", ens_stride); return DABGPU_ERR_INVALID_ARG; }
    if (n_ens == 0) return DABGPU_OK;
    if (n_sub == 0) return dabgpu_fic_decode_ring(c, d_hist, n_ens, ens_stride, d_newest_slot, d_fib_bytes, d_fic_results, tie_rule, stream);
    const fic_request fic = {d_fib_bytes, d_fic_results};
    return msc_decode_any(c, d_hist, n_ens, ens_stride, hist_frames, 0, d_newest_slot, h_sub, n_sub, d_msc_out, out_ens_stride, d_msc_results,
                          tie_rule, stream, bits_layout, &fic);
}

extern "C" int dabgpu_msc_decode_ring(dabgpu_ctx* c, const int8_t* d_hist, size_t n_ens, size_t ens_stride, int hist_frames,
                                      const int32_t* d_newest_slot, const dabgpu_subchannel* h_sub, int n_sub, uint8_t* d_out,
                                      size_t out_ens_stride, dabgpu_codeword_result* d_results, int tie_rule, void* stream) {
    return dabgpu_msc_decode_ring_layout(c, d_hist, n_ens, ens_stride, hist_frames, d_newest_slot, h_sub, n_sub, d_out, out_ens_stride,
                                         d_results, tie_rule, DABGPU_BITS_NATURAL, stream);
}

// ------------------------------------------------------------------------------------------------
// single-stream host-buffer forms (C++ mirror classes)
// ------------------------------------------------------------------------------------------------
static int decode_one_sync(dabgpu_ctx* c, dabgpu_cw_desc D, const int8_t* h_src, size_t n_src, uint8_t* h_out, size_t n_out,
                           dabgpu_codeword_result* h_res, int tie_rule) {
    DABGPU_BIND(c);
    int st;
    int8_t* d_src = nullptr; uint8_t* d_out; dabgpu_codeword_result* d_res; dabgpu_cw_desc* d_desc;
    if (h_src && (st = dabgpu_scratch(c, 14, n_src, (void**)&d_src))) return st;
    // (the kernel writes (n_steps - 6) / 8 bytes whatever part of them the caller wants back)
    const size_t kernel_out = D.n_steps > 6 ? (size_t)(D.n_steps - 6) / 8 : 0;
    if ((st = dabgpu_scratch(c, 15, std::max<size_t>(std::max(n_out, kernel_out), 16), (void**)&d_out))) return st;
    if ((st = dabgpu_scratch(c, 16, sizeof(dabgpu_codeword_result), (void**)&d_res))) return st;
    if ((st = dabgpu_scratch(c, 10, sizeof(dabgpu_cw_desc), (void**)&d_desc))) return st;
    hipStream_t s = c->stream;
    if (h_src) D.d_src = (uint64_t)(uintptr_t)d_src;
    D.d_out = (uint64_t)(uintptr_t)d_out;
    if ((st = dabgpu_host_validate_codeword(D, 0))) return st;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    if (h_src) CK(hipMemcpyAsync(d_src, h_src, n_src, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_desc, &D, sizeof(D), hipMemcpyHostToDevice, s));
    if ((st = run_viterbi(c, d_desc, 1, D.n_steps, D.n_steps > 6 ? (D.n_steps - 6) / 8 : 0, tie_rule, d_res, s))) return st;
    if (n_out) CK(hipMemcpyAsync(h_out, d_out, n_out, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(h_res, d_res, sizeof(*h_res), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
#undef CK
    return DABGPU_OK;
}

extern "C" int dabgpu_fic_decode_group_host_sync(dabgpu_ctx* c, const int8_t* h_bits, uint8_t* h_bytes, uint32_t* crc_ok_mask,
                                                 uint64_t* path_error, int tie_rule) {
    if (!c || !h_bits || !h_bytes) { dabgpu_set_error("fic_decode_group_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_HOST_LOCK(c);
    dabgpu_cw_desc D = {};
    D.n_steps = 774;
    D.seg_pi[0] = 16; D.seg_steps[0] = 32 * 21;
    D.seg_pi[1] = 15; D.seg_steps[1] = 32 * 3;
    D.n_crc_blocks = 3;
    D.d_src = 1;    // placeholder, replaced by the staging buffer
    dabgpu_codeword_result R;
    const int st = decode_one_sync(c, D, h_bits, DABGPU_NB_FIB_GROUP_BITS, h_bytes, 96, &R, tie_rule);
    if (st) return st;
    if (crc_ok_mask) *crc_ok_mask = R.crc_ok_mask;
    if (path_error) *path_error = R.path_error;
    return DABGPU_OK;
}

extern "C" int dabgpu_viterbi_decode_host_sync(dabgpu_ctx* c, const int8_t* h_src, size_t n_src, const uint32_t* seg_pi,
                                               const uint32_t* seg_steps, uint32_t start_state, uint32_t end_state, uint32_t flags,
                                               uint8_t* h_out, size_t n_out_bytes, uint64_t* path_error, int tie_rule) {
    if (!c || !h_src || !seg_pi || !seg_steps || !h_out) { dabgpu_set_error("viterbi_decode_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_HOST_LOCK(c);
    dabgpu_cw_desc D = {};
    uint32_t steps = 0; size_t need = 12;
    for (int k = 0; k < 4; k++) {
        D.seg_pi[k] = seg_steps[k] ? seg_pi[k] : 0; D.seg_steps[k] = seg_steps[k]; steps += seg_steps[k];
        need += (size_t)(seg_steps[k] / 8) * (8 + seg_pi[k]);
    }
    D.n_steps = steps + 6;
    D.start_state = start_state; D.end_state = end_state; D.flags = flags;
    D.d_src = 1;
    if (n_src < need || n_out_bytes * 8 != steps) {
        dabgpu_set_error("viterbi_decode_host_sync: %zu soft bits given, %zu needed; %zu output bytes for %u information bits", n_src, need, n_out_bytes, steps);
        return DABGPU_ERR_INVALID_ARG;
    }
    dabgpu_codeword_result R;
    const int st = decode_one_sync(c, D, h_src, need, h_out, n_out_bytes, &R, tie_rule);
    if (st) return st;
    if (path_error) *path_error = R.path_error;
    return DABGPU_OK;
}

extern "C" int dabgpu_viterbi_decode_depunctured_host_sync(dabgpu_ctx* c, const int8_t* h_mother, size_t n_steps, uint32_t start_state,
                                                           uint32_t end_state, uint8_t* h_out, size_t n_out_bytes, uint64_t* path_error, int tie_rule) {
    if (!c || !h_mother || (!h_out && n_out_bytes)) { dabgpu_set_error("viterbi_decode_depunctured_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG; }
    if (n_steps < 1 || n_steps > DABGPU_MAX_TRELLIS_STEPS) {
        dabgpu_set_error("viterbi_decode_depunctured_host_sync: n_steps %zu out of range (1 .. %u)", n_steps, (unsigned)DABGPU_MAX_TRELLIS_STEPS);
        return DABGPU_ERR_INVALID_ARG;
    }
    if (n_out_bytes && n_out_bytes * 8 + 6 > n_steps) {
        dabgpu_set_error("viterbi_decode_depunctured_host_sync: a trace-back of %zu bytes starts at decision word %zu, only %zu steps were decoded",
                         n_out_bytes, n_out_bytes * 8 + 5, n_steps);
        return DABGPU_ERR_INVALID_ARG;
    }
    DABGPU_HOST_LOCK(c);
    dabgpu_cw_desc D = {};
    D.start_state = start_state; D.end_state = end_state; D.flags = DABGPU_CW_RAW | DABGPU_CW_DEPUNCTURED;
    D.d_src = 1;
    dabgpu_codeword_result R;
    // whole length: the path error (and the bytes, when the trace-back starts at the last step)
    const bool same = n_out_bytes * 8 + 6 == n_steps;
    D.n_steps = (uint32_t)n_steps;
    int st = decode_one_sync(c, D, h_mother, 4 * n_steps, h_out, same ? n_out_bytes : 0, &R, tie_rule);
    if (st) return st;
    if (path_error) *path_error = R.path_error;
    if (same || n_out_bytes == 0) return DABGPU_OK;
    // the trace-back starts earlier: decode the prefix that ends there (same decisions for its steps), from the same end state
    D.n_steps = (uint32_t)(n_out_bytes * 8 + 6);
    return decode_one_sync(c, D, h_mother, 4 * (size_t)D.n_steps, h_out, n_out_bytes, &R, tie_rule);
}

struct dabgpu_msc_stream {
    dabgpu_ctx* ctx;
    dabgpu_subchannel sc;
    dabgpu_cw_desc proto;       // plan with ring geometry, d_src = ring base
    int8_t* d_ring;
    int8_t* d_logical;
    int n_bits;
    int n_out_bytes;
    int next_slot;
    int stored;
    // Consume (push_cif) only files the CIF in this page-locked twin of the ring; a slot crosses to the device when a call that reads the
    // device ring comes (deinterleave_sync / decode_sync).  A decoder whose results come from its demodulator's frame session
    // (dab-radio_amd/host/dab/dabgpu_frame_batcher.h) never reads its own ring: its DecodeCIF then costs a 3 KB host copy, not a DMA.
    int8_t* h_ring;
    uint32_t dirty;             // bit k: slot k of h_ring is newer than the device's
};

// the slots filed since the last device read, uploaded on the context's stream (the callers synchronise with it before they return, so
// h_ring is not overwritten under a copy in flight)
static int msc_stream_flush(dabgpu_msc_stream* s) {
    for (int k = 0; k < 16 && s->dirty; k++) {
        if (!(s->dirty >> k & 1u)) continue;
        const int st = dabgpu_check_hip(hipMemcpyAsync(s->d_ring + (size_t)k * s->n_bits, s->h_ring + (size_t)k * s->n_bits, (size_t)s->n_bits,
                                                       hipMemcpyHostToDevice, s->ctx->stream), "hipMemcpyAsync(msc stream ring)");
        if (st) return st;
        s->dirty &= ~(1u << k);
    }
    return DABGPU_OK;
}

extern "C" int dabgpu_msc_stream_create(dabgpu_ctx* c, const dabgpu_subchannel* sc, dabgpu_msc_stream** out) {
    if (!c || !sc || !out) return DABGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int pi[4], lx[4], nb = 0;
    {   // the same checks as the batch decoders': a valid profile, inside the CIF, consuming no more soft bits than the sub-channel holds
        std::vector<dabgpu_msc_plan> one;
        const int pst = dabgpu_host_build_msc_plans(sc, 1, one, nullptr, nullptr, nullptr);
        if (pst) return pst;
    }
    if (dabgpu_subchannel_plan(sc, pi, lx, &nb) < 0) { dabgpu_set_error("msc_stream_create: invalid protection profile"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    dabgpu_msc_stream* s = new dabgpu_msc_stream();
    s->ctx = c; s->sc = *sc; s->n_bits = sc->length * 64; s->n_out_bytes = nb; s->next_slot = 0; s->stored = 0;
    s->d_ring = nullptr; s->d_logical = nullptr; s->h_ring = nullptr; s->dirty = 0;
    int st = dabgpu_check_hip(hipMalloc((void**)&s->d_ring, (size_t)16 * s->n_bits), "hipMalloc(ring)");
    if (!st) st = dabgpu_check_hip(hipHostMalloc((void**)&s->h_ring, (size_t)16 * s->n_bits, hipHostMallocDefault), "hipHostMalloc(ring)");
    if (!st) st = dabgpu_check_hip(hipMalloc((void**)&s->d_logical, (size_t)s->n_bits), "hipMalloc(logical)");
    // on the context's own stream and waited for: hipMemset runs on the NULL stream, with which a hipStreamNonBlocking stream does not synchronise -- a
    // ring uploaded right after creation (MSC_Decoder creates its stream on its first call-by-call decode) was overwritten by the late zeros
    if (!st) st = dabgpu_check_hip(hipMemsetAsync(s->d_ring, 0, (size_t)16 * s->n_bits, c->stream), "hipMemsetAsync(ring)");
    if (!st) st = dabgpu_check_hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize(ring)");
    if (st) { dabgpu_msc_stream_destroy(s); return st; }
    dabgpu_cw_desc& D = s->proto;
    D = dabgpu_cw_desc{};
    uint32_t steps = 0;
    for (int k = 0; k < 4; k++) { D.seg_pi[k] = lx[k] ? (uint32_t)pi[k] : 0u; D.seg_steps[k] = 32u * (uint32_t)lx[k]; steps += D.seg_steps[k]; }
    D.n_steps = steps + 6;
    D.d_src = (uint64_t)(uintptr_t)s->d_ring;
    D.n_slots = 16; D.cifs_per_frame = 1; D.frame_stride = (uint32_t)s->n_bits; D.cif_stride = 0;
    *out = s;
    return DABGPU_OK;
}

extern "C" void dabgpu_msc_stream_destroy(dabgpu_msc_stream* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    if (s->d_ring) (void)hipFree(s->d_ring);
    if (s->d_logical) (void)hipFree(s->d_logical);
    if (s->h_ring) { (void)hipStreamSynchronize(s->ctx->stream); (void)hipHostFree(s->h_ring); }
    delete s;
}

extern "C" int dabgpu_msc_stream_push_cif(dabgpu_msc_stream* s, const int8_t* h_bits) {
    if (!s || !h_bits) return DABGPU_ERR_INVALID_ARG;
    DABGPU_HOST_LOCK(s->ctx);
    // copied here and now: the caller's span is only valid during DecodeCIF (SURVEY 8b ownership)
    memcpy(s->h_ring + (size_t)s->next_slot * s->n_bits, h_bits, (size_t)s->n_bits);
    s->dirty |= 1u << s->next_slot;
    s->next_slot = (s->next_slot + 1) % 16;                    // cif_deinterleaver.cpp:28-33
    if (s->stored < 16) s->stored++;
    return DABGPU_OK;
}

extern "C" int dabgpu_msc_stream_deinterleave_sync(dabgpu_msc_stream* s, int8_t* h_out) {
    if (!s || !h_out) return DABGPU_ERR_INVALID_ARG;
    DABGPU_HOST_LOCK(s->ctx);
    if (s->stored < 16) return DABGPU_ERR_NOT_READY;           // cif_deinterleaver.cpp:40-42
    DABGPU_BIND(s->ctx);
    hipStream_t q = s->ctx->stream;
    int st = msc_stream_flush(s);
    if (!st) st = dabgpu_check_hip(dabgpu_launch_cif_deinterleave(s->d_ring, s->n_bits, 16, (s->next_slot + 15) % 16, s->d_logical, q),
                              "cif_deinterleave launch");
    if (!st) st = dabgpu_check_hip(hipMemcpyAsync(h_out, s->d_logical, (size_t)s->n_bits, hipMemcpyDeviceToHost, q), "hipMemcpyAsync");
    if (!st) st = dabgpu_check_hip(hipStreamSynchronize(q), "hipStreamSynchronize");
    return st;
}

extern "C" int dabgpu_msc_stream_decode_sync(dabgpu_msc_stream* s, uint8_t* h_out, size_t* n_out, uint64_t* path_error, int tie_rule) {
    if (!s || !h_out || !n_out) return DABGPU_ERR_INVALID_ARG;
    DABGPU_HOST_LOCK(s->ctx);
    *n_out = 0;
    if (s->stored < 16) return DABGPU_ERR_NOT_READY;           // msc_decoder.cpp:60-63
    DABGPU_BIND(s->ctx);
    const int fst = msc_stream_flush(s);
    if (fst) return fst;
    dabgpu_cw_desc D = s->proto;
    D.newest_slot = (uint32_t)((s->next_slot + 15) % 16);
    dabgpu_codeword_result R;
    const int st = decode_one_sync(s->ctx, D, nullptr, 0, h_out, (size_t)s->n_out_bytes, &R, tie_rule);
    if (st) return st;
    *n_out = (size_t)s->n_out_bytes;
    if (path_error) *path_error = R.path_error;
    return DABGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// frame session: ONE batched decode per transmission frame behind the single-stream classes
// ------------------------------------------------------------------------------------------------
// The reference fans a frame out to one FIC runner + one MSC runner per sub-channel (src/basic_radio/basic_radio.cpp:41-65), each
// calling its decoder once per FIB group / CIF: 4 + 4 x sub-channels synchronous round trips per frame behind the mirror classes.
// A session keeps the last 8 frames of soft bits on the device; push_frame copies a frame in once and launches the FIC decode and the
// time de-interleave + Viterbi of every registered sub-channel for its 4 CIFs (the batch entry points above, one ensemble), the
// results return to pinned host slots asynchronously, and the classes pick theirs up by (generation, group / CIF) -- see
// dab-radio_amd/host/dab/dabgpu_frame_batcher.h for when a class may use them.
// (struct dabgpu_frame_session: dabgpu_internal.h -- the receiver pipeline, receiver.hip, pushes frames that are already on the device)

// the result block: [4][96] FIB bytes | [4] FIC results | [4][n_sub] MSC results | [4][cif_out] sub-channel bytes
static size_t session_block_bytes(size_t n_sub, size_t cif_out) {
    return 4 * 96 + (4 + 4 * n_sub) * sizeof(dabgpu_codeword_result) + 4 * cif_out;
}
static void block_pointers(uint8_t* base, size_t n_sub, uint8_t** fib, dabgpu_codeword_result** fres, dabgpu_codeword_result** mres, uint8_t** msc) {
    *fib = base;
    *fres = reinterpret_cast<dabgpu_codeword_result*>(base + 4 * 96);
    *mres = *fres + 4;
    *msc = reinterpret_cast<uint8_t*>(*mres + 4 * n_sub);
}
// (re)allocates the session's device block for n_sub sub-channels of cif_out bytes per CIF; the session's stream must be idle
static int session_layout(dabgpu_frame_session* s, size_t n_sub, size_t cif_out) {
    const size_t need = session_block_bytes(n_sub, cif_out);
    if (need > s->block_bytes || !s->d_block) {
        if (s->d_block) (void)hipFree(s->d_block);
        s->d_block = nullptr; s->block_bytes = 0;
        int st = dabgpu_check_hip(hipMalloc((void**)&s->d_block, need), "hipMalloc(session results)");
        if (st) return st;
        s->block_bytes = need;
    }
    block_pointers(s->d_block, n_sub, &s->d_fib, &s->d_fres, &s->d_mres, &s->d_msc);
    return DABGPU_OK;
}
// the slot's pinned block for the session's current layout
static int slot_block(dabgpu_frame_session* s, dabgpu_frame_session::slot& sl) {
    const size_t need = session_block_bytes(s->subs.size(), s->cif_out);
    if (sl.h_block_cap < need) {
        if (sl.h_block) (void)hipHostFree(sl.h_block);
        sl.h_block = nullptr; sl.h_block_cap = 0;
        int st = dabgpu_check_hip(hipHostMalloc((void**)&sl.h_block, need, hipHostMallocDefault), "hipHostMalloc(session results)");
        if (st) return st;
        sl.h_block_cap = need;
    }
    block_pointers(sl.h_block, s->subs.size(), &sl.h_fib, &sl.h_fres, &sl.h_mres, &sl.h_msc);
    return DABGPU_OK;
}

extern "C" int dabgpu_frame_session_create(dabgpu_frame_session** out, int device) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    *out = nullptr;
    dabgpu_frame_session* s = new dabgpu_frame_session();
    int st = dabgpu_create(&s->ctx, device, nullptr, nullptr);
    if (!st) st = dabgpu_check_hip(hipMalloc((void**)&s->d_hist, (size_t)dabgpu_frame_session::H * DABGPU_NB_FRAME_BITS), "hipMalloc(session history)");
    if (!st) st = dabgpu_check_hip(hipMemsetAsync(s->d_hist, 0, (size_t)dabgpu_frame_session::H * DABGPU_NB_FRAME_BITS, s->ctx->stream), "hipMemsetAsync(session history)");
    if (!st) st = dabgpu_check_hip(hipStreamSynchronize(s->ctx->stream), "hipStreamSynchronize(session history)");      // (not the NULL stream: see dabgpu_msc_stream_create)
    if (!st) st = session_layout(s, 0, 0);
    for (auto& sl : s->slots) {
        if (!st) st = slot_block(s, sl);
        if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&sl.done, dabgpu_wait_event_flags(false)), "hipEventCreate(session)");
        if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&sl.ev_ready, dabgpu_wait_event_flags(false)), "hipEventCreate(session)");
        if (!st) st = dabgpu_check_hip(hipEventCreateWithFlags(&sl.ev_copied, dabgpu_wait_event_flags(false)), "hipEventCreate(session)");
    }
    if (st) { dabgpu_frame_session_destroy(s); return st; }
    *out = s;
    return DABGPU_OK;
}

int dabgpu_frame_session_create_store(dabgpu_frame_session** out, dabgpu_ctx* ctx) {
    dabgpu_frame_session* s = new dabgpu_frame_session();
    s->ctx = ctx;
    s->owns_ctx = false;
    *out = s;
    return DABGPU_OK;
}

extern "C" void dabgpu_frame_session_destroy(dabgpu_frame_session* s) {
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        if (s->owns_ctx) (void)hipStreamSynchronize(s->ctx->stream);
    }
    for (auto& sl : s->slots) {
        if (sl.h_block) (void)hipHostFree(sl.h_block);
        if (sl.h_bits) (void)hipHostFree(sl.h_bits);
        if (sl.h_aux) (void)hipHostFree(sl.h_aux);
        if (sl.h_fft) (void)hipHostFree(sl.h_fft);
        if (sl.h_dq) (void)hipHostFree(sl.h_dq);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.ev_ready) (void)hipEventDestroy(sl.ev_ready);
        if (sl.ev_copied) (void)hipEventDestroy(sl.ev_copied);
    }
    if (s->d_hist) (void)hipFree(s->d_hist);
    if (s->d_block) (void)hipFree(s->d_block);
    if (s->ctx && s->owns_ctx) dabgpu_destroy(s->ctx);
    delete s;
}

extern "C" int dabgpu_frame_session_set_subchannels(dabgpu_frame_session* s, const dabgpu_subchannel* subs, int n) {
    if (!s || n < 0 || n > 64 || (n && !subs)) { dabgpu_set_error("frame_session_set_subchannels: invalid argument"); return DABGPU_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lock(s->mu);
    DABGPU_BIND(s->ctx);
    std::vector<uint32_t> off((size_t)n), nb((size_t)n);
    uint32_t total = 0;
    for (int k = 0; k < n; k++) {
        int pi[4], lx[4], bytes = 0;
        if (dabgpu_subchannel_plan(&subs[k], pi, lx, &bytes) < 0 || subs[k].start_address < 0 || subs[k].start_address + subs[k].length > 864) {
            dabgpu_set_error("frame_session_set_subchannels: sub-channel %d has an invalid protection profile or exceeds 864 CU", k);
            return DABGPU_ERR_INVALID_ARG;
        }
        off[(size_t)k] = total; nb[(size_t)k] = (uint32_t)bytes; total += (uint32_t)bytes;
    }
    int st = dabgpu_check_hip(hipStreamSynchronize(s->ctx->stream), "hipStreamSynchronize(session)");
    if (st) return st;
    if ((st = session_layout(s, (size_t)n, total))) return st;
    s->subs.assign(subs, subs + n); s->sub_off = off; s->sub_n = nb; s->cif_out = total;
    return DABGPU_OK;
}

// The decode of generation `gen` on the session's stream: FIC of the frame in history slot gen % H, MSC of its 4 CIFs (the time
// de-interleaver reaches 4 frames back), results to the slot's pinned buffers, the slot's done event.  s->mu held.
static int session_decode(dabgpu_frame_session* s, uint64_t gen, dabgpu_frame_session::slot& sl, int decode_fic, int tie_rule) {
    dabgpu_ctx* c = s->ctx;
    hipStream_t q = c->stream;
    const int hs = (int)(gen % dabgpu_frame_session::H);
    int8_t* d_frame = s->d_hist + (size_t)hs * DABGPU_NB_FRAME_BITS;
    const int n_sub = (int)s->subs.size();
    int st;
    sl.fic = decode_fic != 0;
    sl.subs = s->subs; sl.sub_off = s->sub_off; sl.sub_n = s->sub_n; sl.cif_out = s->cif_out;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    if ((st = slot_block(s, sl))) return st;
    if (decode_fic && !n_sub) {
        if ((st = dabgpu_fic_decode_frames(c, d_frame, 1, DABGPU_NB_FRAME_BITS, s->d_fib, s->d_fres, tie_rule, q))) return st;
    }
    if (n_sub) {
        const size_t need = (size_t)4 * s->cif_out;
        // the frame's FIC and sub-channels in one call (one launch in the wave mapping: msc_decode_any)
        if (decode_fic) st = dabgpu_decode_frames_layout(c, s->d_hist, 1, (size_t)dabgpu_frame_session::H * DABGPU_NB_FRAME_BITS, dabgpu_frame_session::H, hs,
                                                         s->subs.data(), n_sub, s->d_fib, s->d_fres, s->d_msc, need, s->d_mres, tie_rule, DABGPU_BITS_NATURAL, q);
        else st = dabgpu_msc_decode_frames(c, s->d_hist, 1, (size_t)dabgpu_frame_session::H * DABGPU_NB_FRAME_BITS, dabgpu_frame_session::H, hs,
                                           s->subs.data(), n_sub, s->d_msc, need, s->d_mres, tie_rule, q);
        if (st) return st;
    }
    // one copy: the whole block with sub-channels, its FIC head without
    if (n_sub || decode_fic)
        CK(hipMemcpyAsync(sl.h_block, s->d_block, n_sub ? session_block_bytes((size_t)n_sub, s->cif_out) : session_block_bytes(0, 0), hipMemcpyDeviceToHost, q));
#undef CK
    return DABGPU_OK;
}

extern "C" int dabgpu_frame_session_push_frame(dabgpu_frame_session* s, const int8_t* h_bits, int decode_fic, int tie_rule, uint64_t* generation) {
    if (!s || !h_bits) { dabgpu_set_error("frame_session_push_frame: null argument"); return DABGPU_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lock(s->mu);
    dabgpu_ctx* c = s->ctx;
    DABGPU_BIND(c);
    hipStream_t q = c->stream;
    if (s->next_reserve != s->next_gen) { dabgpu_set_error("frame_session_push_frame: a reserved frame has not been committed"); return DABGPU_ERR_INVALID_ARG; }
    const uint64_t gen = s->next_gen;
    const int hs = (int)(gen % dabgpu_frame_session::H);
    dabgpu_frame_session::slot& sl = s->slots[gen % dabgpu_frame_session::R];
    int st;
    if (sl.pending) {                                                   // the slot's previous frame (R frames ago)
        if ((st = dabgpu_check_hip(hipEventSynchronize(sl.done), "hipEventSynchronize(session)"))) return st;
        sl.pending = false;
    }
    int8_t* d_frame = s->d_hist + (size_t)hs * DABGPU_NB_FRAME_BITS;
    if ((st = dabgpu_stage_h2d(c, d_frame, h_bits, DABGPU_NB_FRAME_BITS, q))) return st;
    sl.gen = ~0ull;
    if ((st = session_decode(s, gen, sl, decode_fic, tie_rule))) return st;
    if ((st = dabgpu_check_hip(hipEventRecord(sl.done, q), "hipEventRecord(session)"))) return st;
    sl.pending = true;
    sl.gen = gen;
    s->next_gen = gen + 1;
    s->next_reserve = gen + 1;
    if (generation) *generation = gen;
    return DABGPU_OK;
}

int dabgpu_session_reserve(dabgpu_frame_session* s, hipStream_t producer, uint64_t* gen_out, int8_t** d_frame_bits, dabgpu_frame_session::slot** slot_out) {
    DABGPU_BIND(s->ctx);
    int st;
    {   // the slot's previous frame (R frames ago) is waited for OUTSIDE the session's lock: the decode thread's commit and every decoder's fetch of
        // this receiver (and, through the batcher, of the others) take that lock.  Reservations come from one thread (the header's contract),
        // so next_reserve cannot move while this one waits
        hipEvent_t done = nullptr;
        uint64_t gen0;
        {
            std::lock_guard<std::mutex> lock(s->mu);
            gen0 = s->next_reserve;
            dabgpu_frame_session::slot& sl0 = s->slots[gen0 % dabgpu_frame_session::R];
            if (sl0.pending) done = sl0.done;
        }
        if (done) {
            if ((st = dabgpu_check_hip(hipEventSynchronize(done), "hipEventSynchronize(session)"))) return st;
            std::lock_guard<std::mutex> lock(s->mu);
            dabgpu_frame_session::slot& sl0 = s->slots[gen0 % dabgpu_frame_session::R];
            if (s->next_reserve == gen0 && sl0.done == done) sl0.pending = false;
        }
    }
    std::lock_guard<std::mutex> lock(s->mu);
    const uint64_t gen = s->next_reserve;
    // the commits (decode enqueued) may lag the reservations by a few frames (dabgpu_receiver_submit_demod / _submit_decode on two threads), but the
    // result slot and the history slot of `gen` must be free: R - 1 frames at most may be reserved and not yet waited for
    if (gen >= s->next_gen + (uint64_t)(dabgpu_frame_session::R - 1)) {
        dabgpu_set_error("session_reserve: %d frames reserved and not committed (at most %d)", (int)(gen - s->next_gen), dabgpu_frame_session::R - 1);
        return DABGPU_ERR_NOT_READY;
    }
    dabgpu_frame_session::slot& sl = s->slots[gen % dabgpu_frame_session::R];
    if (sl.pending) {                                                  // (only if the wait above raced with a commit of that slot: cannot happen with one reserving thread)
        if ((st = dabgpu_check_hip(hipEventSynchronize(sl.done), "hipEventSynchronize(session)"))) return st;
        sl.pending = false;
    }
    // history slot gen % H holds frame gen - H, which the decodes of the frames gen - H .. gen - 4 read (5 frames = 16 CIFs + the frame's own
    // 4): the producer may overwrite it once the decode of frame gen - 4 has run (the decodes run in order on one stream)
    if (gen >= 4) {
        if (gen - 4 >= s->next_gen) {
            dabgpu_set_error("session_reserve: the decode of frame %llu has not been submitted, frame %llu would overwrite what it reads (at most 4 frames "
                             "between dabgpu_receiver_submit_demod and dabgpu_receiver_submit_decode)", (unsigned long long)(gen - 4), (unsigned long long)gen);
            return DABGPU_ERR_NOT_READY;
        }
        dabgpu_frame_session::slot& old = s->slots[(gen - 4) % dabgpu_frame_session::R];
        if (old.pending && old.gen == gen - 4 &&
            (st = dabgpu_check_hip(hipStreamWaitEvent(producer, old.done, 0), "hipStreamWaitEvent(session history)"))) return st;
    }
    sl.gen = ~0ull;
    s->next_reserve = gen + 1;
    *gen_out = gen;
    *d_frame_bits = s->d_hist + (size_t)(gen % dabgpu_frame_session::H) * DABGPU_NB_FRAME_BITS;
    *slot_out = &sl;
    return DABGPU_OK;
}

int dabgpu_session_commit(dabgpu_frame_session* s, uint64_t gen, hipEvent_t ready, size_t bits_bytes, int decode, int decode_fic, int tie_rule,
                          hipEvent_t producer_done) {
    std::lock_guard<std::mutex> lock(s->mu);
    dabgpu_ctx* c = s->ctx;
    DABGPU_BIND(c);
    if (gen != s->next_gen) { dabgpu_set_error("session_commit: generation %llu was not the one reserved", (unsigned long long)gen); return DABGPU_ERR_INVALID_ARG; }
    if (gen >= s->next_reserve) {                                       // (a decode submitted before its demodulation: the slot's ready event is stale or was never recorded)
        dabgpu_set_error("session_commit: generation %llu has not been reserved (dabgpu_receiver_submit_demod comes first)", (unsigned long long)gen);
        return DABGPU_ERR_NOT_READY;
    }
    hipStream_t q = c->stream;
    dabgpu_frame_session::slot& sl = s->slots[gen % dabgpu_frame_session::R];
    int st;
    if (ready && (st = dabgpu_check_hip(hipStreamWaitEvent(q, ready, 0), "hipStreamWaitEvent(session producer)"))) return st;
    if (bits_bytes) {
        if (!sl.h_bits && (st = dabgpu_check_hip(hipHostMalloc((void**)&sl.h_bits, DABGPU_NB_FRAME_BITS, hipHostMallocDefault), "hipHostMalloc(session bits)"))) return st;
        if ((st = dabgpu_check_hip(hipMemcpyAsync(sl.h_bits, s->d_hist + (size_t)(gen % dabgpu_frame_session::H) * DABGPU_NB_FRAME_BITS, bits_bytes,
                                                  hipMemcpyDeviceToHost, q), "hipMemcpyAsync(session bits)"))) return st;
    }
    sl.fic = false; sl.subs.clear(); sl.sub_off.clear(); sl.sub_n.clear(); sl.cif_out = 0;
    if (decode && (st = session_decode(s, gen, sl, decode_fic, tie_rule))) return st;
    if (producer_done && (st = dabgpu_check_hip(hipStreamWaitEvent(q, producer_done, 0), "hipStreamWaitEvent(session producer copies)"))) return st;
    if ((st = dabgpu_check_hip(hipEventRecord(sl.done, q), "hipEventRecord(session)"))) return st;
    sl.pending = true;
    sl.gen = gen;
    s->next_gen = gen + 1;
    return DABGPU_OK;
}

// a reservation whose producer failed before anything of the frame was enqueued for the decoder: give the generation back (only the NEWEST
// reservation can be returned, and only while it has not been committed)
void dabgpu_session_unreserve(dabgpu_frame_session* s, uint64_t gen) {
    std::lock_guard<std::mutex> lock(s->mu);
    if (gen + 1 == s->next_reserve && gen >= s->next_gen) s->next_reserve = gen;
}

int dabgpu_session_slot(dabgpu_frame_session* s, uint64_t gen, dabgpu_frame_session::slot** out) {
    dabgpu_frame_session::slot& sl = s->slots[gen % dabgpu_frame_session::R];
    if (sl.gen != gen) return DABGPU_ERR_NOT_READY;                     // never pushed, or overwritten by a later frame
    if (sl.pending) {
        DABGPU_BIND(s->ctx);
        int st = dabgpu_check_hip(hipEventSynchronize(sl.done), "hipEventSynchronize(session)");
        if (st) return st;
        sl.pending = false;
    }
    *out = &sl;
    return DABGPU_OK;
}
static int session_slot(dabgpu_frame_session* s, uint64_t gen, dabgpu_frame_session::slot** out) { return dabgpu_session_slot(s, gen, out); }

extern "C" int dabgpu_frame_session_fetch_fib_group(dabgpu_frame_session* s, uint64_t generation, int group, uint8_t* h_bytes,
                                                    uint32_t* crc_ok_mask, uint64_t* path_error) {
    if (!s || group < 0 || group > 3 || !h_bytes) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(s->mu);
    dabgpu_frame_session::slot* sl = nullptr;
    int st = session_slot(s, generation, &sl);
    if (st) return st;
    if (!sl->fic) return DABGPU_ERR_NOT_READY;
    memcpy(h_bytes, sl->h_fib + 96 * group, 96);
    if (crc_ok_mask) *crc_ok_mask = sl->h_fres[group].crc_ok_mask;
    if (path_error) *path_error = sl->h_fres[group].path_error;
    return DABGPU_OK;
}

extern "C" int dabgpu_frame_session_fetch_cif(dabgpu_frame_session* s, uint64_t generation, const dabgpu_subchannel* sc, int cif,
                                              uint8_t* h_bytes, size_t capacity, size_t* n_bytes, uint64_t* path_error) {
    if (!s || !sc || cif < 0 || cif > 3 || !h_bytes || !n_bytes) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(s->mu);
    dabgpu_frame_session::slot* sl = nullptr;
    int st = session_slot(s, generation, &sl);
    if (st) return st;
    for (size_t k = 0; k < sl->subs.size(); k++) {
        if (memcmp(&sl->subs[k], sc, sizeof(dabgpu_subchannel)) != 0) continue;
        if (sl->sub_n[k] > capacity) return DABGPU_ERR_INVALID_ARG;
        memcpy(h_bytes, sl->h_msc + (size_t)cif * sl->cif_out + sl->sub_off[k], sl->sub_n[k]);
        *n_bytes = sl->sub_n[k];
        if (path_error) *path_error = sl->h_mres[(size_t)cif * sl->subs.size() + k].path_error;
        return DABGPU_OK;
    }
    return DABGPU_ERR_NOT_READY;                                        // the sub-channel was not registered when that frame was pushed
}
