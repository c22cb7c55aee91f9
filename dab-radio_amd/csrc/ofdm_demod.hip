// ofdm_demod.hip -- gfx950 kernels for the OFDM half of the hot path.
//
// ofdm_demod_kernel: one 256-thread workgroup walks a run of consecutive OFDM symbols of one frame.
// Per symbol, entirely on-chip between one coalesced HBM read of the 2552 IQ samples (16 B per lane)
// and one coalesced HBM write of the 3072 soft bits (16 B per lane):
//   PLL (ref: src/ofdm/dsp/apply_pll.cpp:81-117, AVX+FMA arithmetic)            -> registers
//   cyclic-prefix correlation (ref: ofdm_demodulator.cpp:768-777)               -> wave shuffle tree
//   2048-pt FFT = 4x8x8x8 butterflies (replaces FFTW, ofdm_demodulator.cpp:891-894)
//        exchange 1 crosses waves (LDS + one __syncthreads); exchanges 2 and 3 are 8x8 transposes
//        between lane-index bits and the register index inside a wave (wave-private LDS patch)
//   DQPSK against the previous symbol kept in registers (ref: :842-865)
//   L-inf normalise, x(-127), truncate to int8 (ref: :57-72, :867-889)
//   frequency de-interleave (ref: :874) as a byte scatter into LDS, then 16-byte row stores.
//
// The arithmetic contract (operation order, explicit FMAs, reduction tree) is the one DESIGN.md
// section 3 states; this file is compiled with -ffp-contract=off -fno-slp-vectorize so only the FMAs
// written here exist and nothing is re-associated.  Complex values are kept as two scalar floats on
// purpose: on gfx950 a packed fp32 op (v_pk_*) issues at half the rate of a scalar one (measured with
// tools/ubench/valu_rate.hip), so packing buys no ALU throughput and costs operand-pairing moves and
// hazard nops (A/B: 0.51 ms packed vs 0.46 ms scalar per 1024 frames, profiles/r01/README.md).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dabgpu_internal.h"
#include "iq_decode.h"
#include "ofdm_device.h"

// wave priority by phase of the symbol loop, one hex digit per point (F = leave as it is): start of symbol, before the
// correlation, after barrier 1, after barrier 2, after the last radix-8 pass, end of the demapper
constexpr int DEMOD_PRIO = 0x0F3FFF;       // measured choice (profiles/r02/ab_notes.md): 3 from the exchange barrier to the end of the demapper, 0 for PLL + radix 4
#define PRIO_AT(k) do { constexpr int prio_ = (DEMOD_PRIO >> (4 * (k))) & 0xF; if (prio_ != 0xF) __builtin_amdgcn_s_setprio(prio_); } while (0)

namespace dabgpu {

// Input sample formats the loader can read straight from HBM (anything else goes through iq_convert_kernel first).
// The decode is iq_decode.h's, i.e. the same operations as the stand-alone converter: results are bit-identical to
// convert-then-demodulate while the kernel reads 2 or 4 bytes per sample instead of 8.
enum { SRC_C32 = 0, SRC_U8 = 1, SRC_S8 = 2, SRC_S16 = 3 };
template <int SRC> struct src_bytes { static constexpr int value = (SRC == SRC_C32) ? 8 : (SRC == SRC_S16) ? 4 : 2; };

// two consecutive IQ samples of a frame -> (re0, im0, re1, im1), read through a buffer descriptor of the frame:
// voffset = this lane's loop-invariant byte offset, soffset = the uniform byte
// offset of (symbol, slot) -- all per-symbol address arithmetic is scalar, none is left on the vector ALU, which is the unit this
// kernel is bound by (profiles/r02/ab_notes.md).  Loads past the frame return zero.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
constexpr int BUF_NT = 2;                  // cache-policy operand: non-temporal
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int SRC>
__device__ __forceinline__ f4 load_pair_buf(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes) {
    if constexpr (SRC == SRC_C32) {
        const u4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, (int)uniform_bytes, BUF_NT);   // streamed once (-2 %)
        return f4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
    } else if constexpr (SRC == SRC_S16) {
        raw_words<2> r;
        const u2v q = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_bytes, (int)uniform_bytes, 0);
        r.w[0] = q.x; r.w[1] = q.y;
        return f4{decode<K_S16, 2, false>(r, 0), decode<K_S16, 2, false>(r, 1), decode<K_S16, 2, false>(r, 2), decode<K_S16, 2, false>(r, 3)};
    } else {
        constexpr comp_kind K = (SRC == SRC_U8) ? K_U8 : K_S8;
        raw_words<1> r;
        r.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_bytes, (int)uniform_bytes, 0);
        return f4{decode<K, 1, false>(r, 0), decode<K, 1, false>(r, 1), decode<K, 1, false>(r, 2), decode<K, 1, false>(r, 3)};
    }
}

// blockIdx -> work unit so that consecutive units (chunks of one frame share a halo symbol) sit on one XCD
// (blocks are dealt round-robin to the 8 XCDs, each with a private L2)
__device__ __forceinline__ int xcd_remap(int b, int G) {
    const int nx = 8, q = G / nx, r = G % nx, x = b % nx, s = b / nx;
    return (x < r) ? (x * (q + 1) + s) : (r * (q + 1) + (x - r) * q + s);
}

// In-place decimation-in-frequency view of the 4x8x8x8 factorisation: after the radix-4 pass the transform splits
// into four independent 512-point problems, one per wave.  Thread (w, l) ends holding bins
//   Kb + 256*k,  Kb = w + 4*(l>>3) + 32*(l&7),  k = 0..7
// of which k in {0,1,2,5,6,7} are data carriers (thread 0: DC is replaced by bin 768, its k = 3).
// stream banks (ofdm_stream.hip): samples [0, split) of a frame come from the stream's frame buffer, the rest straight
// from the caller's block (no assembly copy); split is even, the block side is only 8-byte aligned
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));

// stream banks: head of the frame from the (complex float) frame buffer, tail from the caller's block in its capture format SRC
// (tail_base is pre-biased so that frame sample n sits at tail_base + n * bytes per sample; only naturally 2/4/8-byte aligned)
// (retained blocks: frame samples [cd, ce) -- even bounds inside [0, split) -- come from the stream's PREVIOUS block, pre-biased like the
// tail; ce = 0 when nothing is carried)
template <int SRC>
__device__ __forceinline__ f4 load_pair_bank(const uint8_t* __restrict__ frame_base, const uint8_t* __restrict__ tail_base, int split, size_t n,
                                             const uint8_t* __restrict__ prev_base, int cd, int ce) {
    const bool head = (int)n < split;
    const bool carried = (int)n >= cd && (int)n < ce;
    if constexpr (SRC == SRC_C32) {
        const uint8_t* p = (head ? (carried ? prev_base : frame_base) : tail_base) + n * 8;
        const f4u v = *reinterpret_cast<const f4u*>(p);
        return f4{v.x, v.y, v.z, v.w};
    } else {
        if (head && !carried) return *reinterpret_cast<const f4*>(frame_base + n * 8);
        const uint8_t* p = (head ? prev_base : tail_base) + n * src_bytes<SRC>::value;
        if constexpr (SRC == SRC_S16) {
            typedef uint32_t u32a4 __attribute__((aligned(4)));
            raw_words<2> r;
            r.w[0] = reinterpret_cast<const u32a4*>(p)[0]; r.w[1] = reinterpret_cast<const u32a4*>(p)[1];
            return f4{decode<K_S16, 2, false>(r, 0), decode<K_S16, 2, false>(r, 1), decode<K_S16, 2, false>(r, 2), decode<K_S16, 2, false>(r, 3)};
        } else {
            constexpr comp_kind K = (SRC == SRC_U8) ? K_U8 : K_S8;
            typedef uint32_t u32a2 __attribute__((aligned(2)));
            raw_words<1> r;
            r.w[0] = *reinterpret_cast<const u32a2*>(p);
            return f4{decode<K, 1, false>(r, 0), decode<K, 1, false>(r, 1), decode<K, 1, false>(r, 2), decode<K, 1, false>(r, 3)};
        }
    }
}

// optional fused tail of the kernel = ofdm_phase_kernel of the frame (total_phase / fine_freq may each be null);
// frame k's fine-frequency word is fine_freq[k * fine_stride] (1: a plain array; 6: the freq_fine members of dabgpu_sync_state records)
struct demod_phase_tail { float* total_phase; float* fine_freq; float beta; int fine_stride; };

// where a frame's samples start (frame-aligned batches; stream banks use dabgpu_frame_desc instead): frame k begins `stride` samples
// after frame k - 1.  sync != nullptr (dabgpu_ofdm_sync_demod_frames): the frame of receiver k begins prs_offset + fine_time_offset
// samples into its slice, is skipped when the impulse-peak test failed (the caller resets that receiver, ofdm_demodulator.cpp:529-532)
// and is corrected by freq_coarse + freq_fine (:672) -- the record ofdm_sync_kernel has just written, read on the device
struct demod_frame_src { size_t stride; const dabgpu_sync_state* sync; int prs_offset; };

// LDS of one workgroup (float2 elements unless noted)
constexpr int LDS_TW1 = 6 * 256;           // pass-1 twiddles [thread][6]: a private 48-byte slot per thread (registers parked in LDS)
constexpr int LDS_TW3 = 7 * 8;             // pass-3 twiddles [k][lane & 7] (stream-bank and class-order instantiations only: they need the registers)
constexpr int LDS_PHASE = 80 + 40;         // fused phase tail: the frame's 76 correlations (f2) + their 76 angles (float)
constexpr size_t DEMOD_LDS_BYTES = (4 * WAVE_PATCH) * sizeof(f2) + NB_SYM_BITS + 8 * sizeof(f2) + (LDS_TW1 + LDS_TW3 + LDS_PHASE) * sizeof(f2);

// VIEWS = false: the instantiation for callers that want soft bits only (fft_out / dqpsk_out are GUI views of the reference's
// GetFrameFFT() / GetFrameDataVec()): their stores and per-carrier branches leave the symbol loop
//
// Software pipeline (round 2): a symbol's samples are dead once the PLL and the radix-4 pass have consumed them, so the NEXT
// symbol's five 16-byte loads are issued right there, into the same registers, and have the three radix-8 passes, the demapper
// and the end-of-symbol barrier to arrive (no second register set, no LDS staging).  To stay inside 128 VGPRs (4 workgroups
// per CU) the pass-2 twiddles (14 registers) are read from a small LDS table instead.  The vector-memory counter
// of gfx9 is in order, therefore the previous symbol's soft-bit store is issued BEFORE the prefetch (from the middle of the
// next symbol), never between a prefetch and its use.  0.438 -> 0.426 ms per 1024 frames (profiles/r02/ab_notes.md).
//
// CLASSED = true (soft bits only): the MSC symbols leave in time-interleaver class order -- inside every CIF row
// of 55296 soft bits, bit i is stored at (i mod 16) * 3456 + i / 16 -- so that the channel decoder's gather, which needs for one
// output CIF the bits of class c from the CIF that is 15 - bitrev4(c) CIFs old (cif_deinterleaver.cpp:57-68), reads each history row
// in contiguous pieces instead of one byte in sixteen.  The permutation rides on the frequency de-interleave scatter (a second
// set of LDS positions) and on the row store's addresses: no extra pass, no extra instructions.  The FIC symbols stay as they are.
template <int SRC, bool BANK, bool VIEWS = true, bool CLASSED = false>
__global__ __launch_bounds__(256, (VIEWS || (BANK && CLASSED)) ? 3 : 4)      // (the display views, and the stream-bank loader with the
                                                                            // second position set, need a few more registers: 3 workgroups per CU instead of spills)
void ofdm_demod_kernel(const void* __restrict__ iq, const float* __restrict__ freq_offset,
                       int8_t* __restrict__ bits, f2* __restrict__ cp_corr, f2* __restrict__ fft_out_,
                       f2* __restrict__ dqpsk_out_, const f2* __restrict__ tw, const uint16_t* __restrict__ inv_map,
                       int n_frames, int sym_per_chunk, int chunks_per_frame, size_t bits_frame_stride,
                       const dabgpu_frame_desc* __restrict__ desc, const void* __restrict__ tail, size_t tail_stride, demod_phase_tail pt,
                       const void* __restrict__ prev_tail, demod_frame_src fs)
{
    f2* const fft_out = VIEWS ? fft_out_ : nullptr;
    f2* const dqpsk_out = VIEWS ? dqpsk_out_ : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // bufA: the radix-4 outputs, position-indexed, as four 512-element blocks (one per wave's 512-point problem) placed
    // WAVE_PATCH = 576 apart; a wave's transpose patch ALIASES its own block: the wave has read its 8 inputs per lane before
    // it writes the patch (one wave's LDS instructions execute in order) and nobody else touches the block until the
    // end-of-symbol barrier
    f2* bufA = reinterpret_cast<f2*>(smem);                              // 4 x 576 x 8 B
    f2* patch0 = bufA;
    int8_t* obuf = reinterpret_cast<int8_t*>(bufA + 4 * WAVE_PATCH);     // 3072 B
    f2* red = reinterpret_cast<f2*>(obuf + NB_SYM_BITS);                 // 2 x 4 x 8 B
    f2* tw1l = red + 8;                                                  // 256 x 6 x 8 B
    f2* tw3l = tw1l + LDS_TW1;                                           // 7 x 8 x 8 B
    f2* ph_corr = tw3l + LDS_TW3;                                        // 80 x 8 B + 80 x 4 B (fused phase tail)
    // fused phase tail (the launcher asks for it only when this workgroup walks the whole frame): ofdm_phase_kernel's work at the
    // end of the run -- the per-symbol angles in parallel, their sum in symbol order, the fine-frequency update
    const bool phase_tail = !VIEWS && !BANK && (pt.total_phase != nullptr || pt.fine_freq != nullptr);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int unit = xcd_remap(blockIdx.x, gridDim.x);
    const int frame = unit / chunks_per_frame;
    const int chunk = unit % chunks_per_frame;
    if (frame >= n_frames) return;
    // stream banks: frame = stream index; desc[stream].slot = output slot of its completed frame, < 0 = nothing to do
    size_t out_frame = (size_t)frame;
    int split = 0, cd = 0, ce = 0;
    const uint8_t *tbase = nullptr, *pbase = nullptr;
    if constexpr (BANK) {
        const dabgpu_frame_desc d = desc[frame];
        if (d.slot < 0) return;
        out_frame = (size_t)d.slot;
        split = d.split;
        tbase = static_cast<const uint8_t*>(tail) + ((long long)frame * (long long)tail_stride + d.tail_off - (long long)d.split) * src_bytes<SRC>::value;
        if (prev_tail != nullptr && d.carry_end > d.carry_dst) {     // part of the head still sits in the stream's previous block
            cd = d.carry_dst; ce = d.carry_end;
            pbase = static_cast<const uint8_t*>(prev_tail) + ((long long)frame * (long long)tail_stride + d.carry_off) * src_bytes<SRC>::value;
        }
    }

    // DQPSK outputs [out0, out1) need FFTs of symbols [out0, out1]; the last chunk also owns the
    // correlation of symbol 75 and (only when fft_out != nullptr) the display-only NULL symbol 76.
    const int out0 = chunk * sym_per_chunk;
    const int out1 = min(out0 + sym_per_chunk, NB_FRAME_SYMBOLS - 1);
    const bool last_chunk = (out1 == NB_FRAME_SYMBOLS - 1);
    const int sym_end = (last_chunk && fft_out != nullptr) ? NB_FRAME_SYMBOLS : out1;   // inclusive

    float f = freq_offset ? freq_offset[frame] : 0.0f;
    // (stream banks: iq = the complex-float frame buffers, SRC describes the tail source only)
    const uint8_t* fbase = static_cast<const uint8_t*>(iq) + (size_t)frame * (BANK ? (size_t)NB_FRAME_SAMPLES * 8 : fs.stride * src_bytes<SRC>::value);
    if constexpr (!BANK) {
        if (fs.sync != nullptr) {                                            // (uniform; scalar loads)
            const dabgpu_sync_state st = fs.sync[frame];
            if (!st.sync_valid) return;
            f = st.freq_coarse + st.freq_fine;                               // ofdm_demodulator.cpp:672
            fbase += ((long long)fs.prs_offset + (long long)st.fine_time_offset) * (long long)src_bytes<SRC>::value;   // :536-545
        }
    }

    constexpr unsigned SB = BANK ? 8u : (unsigned)src_bytes<SRC>::value;                              // bytes per sample of the frame
    const unsigned lane_off = 2u * SB * (unsigned)t, head_off = 2u * SB * (unsigned)((t >= 4) ? t - 4 : 0);   // byte offsets of this lane
    const __amdgpu_buffer_rsrc_t iq_rs = frame_rsrc(fbase, NB_FRAME_SAMPLES * SB);
    // coalesced loads of one symbol: 16 B per lane, 4 for the FFT body + 1 for the cyclic-prefix head (threads 0..3 have no head
    // sample: they load a valid address and never use it, so that the load stays unconditional inside the wave)
    auto load_symbol = [&](int i, f4 (&v)[4], f4& h) __attribute__((always_inline)) {
        const size_t sym = (size_t)i * NB_SYMBOL_PERIOD;
        const bool dc = (i < NB_FRAME_SYMBOLS) && (i < out1 || i == NB_FRAME_SYMBOLS - 1);
        if constexpr (BANK) {
            // A symbol lies in ONE of the frame's sources -- frame buffer, previous block, current block -- except the one or two that
            // hold a boundary: the choice is uniform, so the symbol is loaded through a buffer descriptor of that source like an aligned
            // frame's (scalar symbol offset + loop-invariant lane offset); only a boundary symbol selects per lane.
            const int lo = (int)sym, hi = lo + NB_SYMBOL_PERIOD;
            const bool in_tail = lo >= split, in_prev = lo >= cd && hi <= ce, in_fb = hi <= split && (hi <= cd || lo >= ce);
            if (in_fb) {
                const __amdgpu_buffer_rsrc_t rs = frame_rsrc(fbase, NB_FRAME_SAMPLES * 8u);
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = load_pair_buf<SRC_C32>(rs, lane_off, (unsigned)(sym + NB_CP + 512 * k) * 8u);
                if (dc) h = load_pair_buf<SRC_C32>(rs, head_off, (unsigned)sym * 8u);
            } else if (in_tail || in_prev) {
                constexpr unsigned SS = (unsigned)src_bytes<SRC>::value;
                const __amdgpu_buffer_rsrc_t rs = frame_rsrc(in_tail ? tbase : pbase, NB_FRAME_SAMPLES * SS);
                const unsigned lane_s = 2u * SS * (unsigned)t, head_s = 2u * SS * (unsigned)((t >= 4) ? t - 4 : 0);
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = load_pair_buf<SRC>(rs, lane_s, (unsigned)(sym + NB_CP + 512 * k) * SS);
                if (dc) h = load_pair_buf<SRC>(rs, head_s, (unsigned)sym * SS);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = load_pair_bank<SRC>(fbase, tbase, split, sym + NB_CP + 2 * t + 512 * k, pbase, cd, ce);
                if (dc) h = load_pair_bank<SRC>(fbase, tbase, split, sym + 2 * ((t >= 4) ? t - 4 : 0), pbase, cd, ce);   // uniform per workgroup
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = load_pair_buf<SRC>(iq_rs, lane_off, (unsigned)(sym + NB_CP + 512 * k) * SB);
            if (dc) h = load_pair_buf<SRC>(iq_rs, head_off, (unsigned)sym * SB);
        }
    };
    // the first symbol's samples are requested before anything else: the table reads and the barrier of the set-up below run while
    // they are on their way (they were issued behind them: two memory latencies in series at the start of every workgroup)
    f4 v[4], h = f4{0.0f, 0.0f, 0.0f, 0.0f};
    load_symbol(out0, v, h);

    // PLL constants: this thread always touches sample pairs (n, n+1) with n & 3 == 2*(t&1)
    const int k0 = 2 * (t & 1);
    const float ss0 = (float)k0 * f, ss1 = (float)(k0 + 1) * f;          // apply_pll.cpp:95-99
    const f2 step0 = mk2(ss0 + 0.25f, ss0), step1 = mk2(ss1 + 0.25f, ss1);
    float gf[5];                                                         // float(i4) * f of the five 16-byte slots: the product
#pragma unroll                                                           // of apply_pll.cpp:103 depends on (thread, slot) only
    for (int k = 0; k < 4; k++) gf[k] = (float)((NB_CP + 2 * t + 512 * k) & ~3) * f;
    gf[4] = (float)((2 * (t - 4)) & ~3) * f;

    // twiddles tw[m] = (cos, -sin)(2 pi m / 2048).  Pass 2 w_512^{lane k} and pass 3 w_64^{(lane&7) k} stay in registers for the
    // whole run of symbols: they are used between two LDS round trips of the transform, where a table read would add a third.
    // The six pass-1 twiddles w_2048^{p k} (p = 2t, 2t+1; k = 1..3) are parked in a private LDS slot of the thread and read back
    // at the start of every symbol: the PLL in front of their use hides the latency.  Same table entries either way.
#pragma unroll
    for (int k = 1; k < 4; k++) { tw1l[6 * t + 2 * (k - 1)] = tw[(2 * t) * k]; tw1l[6 * t + 2 * (k - 1) + 1] = tw[(2 * t + 1) * k]; }
    f2 w2[7];
#pragma unroll
    for (int k = 1; k < 8; k++) w2[k - 1] = tw[4 * lane * k];
    f2 w3[7];
    constexpr bool W3_LDS = BANK || CLASSED;     // (class order keeps a second set of scatter positions instead)
    if constexpr (W3_LDS) {
        if (t < LDS_TW3) tw3l[t] = tw[32 * (t & 7) * ((t >> 3) + 1)];
    } else {
#pragma unroll
        for (int k = 1; k < 8; k++) w3[k - 1] = tw[32 * (lane & 7) * k];
    }
    __syncthreads();
    const f4* const w1p = reinterpret_cast<const f4*>(tw1l + 6 * t);      // (w1a[k], w1b[k]) = w1p[k - 1]
    const f2* const w3p = tw3l + (lane & 7);      // w3[k] = w3p[8 (k - 1)]

    // LDS addresses (float2 element indices); every per-k term below is an instruction immediate
    f2* patch = patch0 + wave * WAVE_PATCH;
    const int la = lane & 7, lb = lane >> 3;
    const int rd2 = lane + WAVE_PATCH * wave;     // pass-2 inputs: + 64 j           (stride-1 across lanes)
    const int ta_w = lane;                        // transpose A write: + 72 r       (element a + 8b + 72r)
    const int ta_r = la + 72 * lb;                // transpose A read : + 8 r'       (element a + 8r' + 72b')
    const int tb_w = la + 72 * lb;                // transpose B write: + 9 r        (element a + 9r + 72b)
    const int tb_r = 9 * la + 72 * lb;            // transpose B read : + r'         (element r' + 9a' + 72b)

    // de-interleave positions of this thread's six active bins Kb + 256*{0,1,2,5,6,7}
    const int Kb = wave + 4 * lb + 32 * la;
    int pos[6];
    pos[0] = inv_map[(Kb == 0) ? 1535 : (767 + Kb)];
    pos[1] = inv_map[1023 + Kb];
    pos[2] = inv_map[1279 + Kb];
    pos[3] = inv_map[Kb];
    pos[4] = inv_map[256 + Kb];
    pos[5] = inv_map[512 + Kb];
    int posc[6];                                  // the same positions in class order: (b mod 16) * 192 + b / 16; the imaginary half sits 96 further
#pragma unroll
    for (int k = 0; k < 6; k++) posc[k] = CLASSED ? ((pos[k] & 15) * 192 + (pos[k] >> 4)) : 0;

    f2 prev[6], keep[6];
#pragma unroll
    for (int k = 0; k < 6; k++) prev[k] = mk2(0.0f, 0.0f);

    const __amdgpu_buffer_rsrc_t bits_rs = frame_rsrc(bits + out_frame * bits_frame_stride, (NB_FRAME_SYMBOLS - 1) * NB_SYM_BITS);
    // the 3072 soft bits of data symbol `row` + 1 sit de-interleaved in obuf: 192 lanes x 16-byte stores (LDS read and HBM store are
    // separate steps so that the read can be issued ahead of the radix-4 exchange writes and the store behind them)
    auto row_read = [&]() __attribute__((always_inline)) -> u4v {
        const uint4 o = reinterpret_cast<const uint4*>(obuf)[(t < NB_SYM_BITS / 16) ? t : 0];
        return u4v{o.x, o.y, o.z, o.w};
    };
    const int row_voff_c = (t / 12) * (NB_CIF_BITS / 16) + 16 * (t % 12);      // CLASSED: class t / 12, bytes 16 (t mod 12) .. + 15 of this symbol's 192
    auto row_write = [&](const int row, const u4v o) __attribute__((always_inline)) {
        if (t < NB_SYM_BITS / 16) {
            if (CLASSED && row >= NB_FIC_SYMBOLS) {         // (uniform) symbol s of CIF q: 192 bytes per class
                const int q = (row - NB_FIC_SYMBOLS) / NB_CIF_SYMBOLS, sy = (row - NB_FIC_SYMBOLS) % NB_CIF_SYMBOLS;
                __builtin_amdgcn_raw_buffer_store_b128(o, bits_rs, row_voff_c, NB_FIC_SYMBOLS * NB_SYM_BITS + q * NB_CIF_BITS + sy * (NB_SYM_BITS / 16), BUF_NT);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(o, bits_rs, 16 * t, row * NB_SYM_BITS, BUF_NT);
            }
        }
    };
    auto store_row = [&](const int row) __attribute__((always_inline)) { row_write(row, row_read()); };

    // one symbol; pv = the six active bins of symbol i - 1 (in), cur = those of symbol i (out)
    auto symbol = [&](const int i, const f2 (&pv)[6], f2 (&cur)[6]) __attribute__((always_inline)) {
        const float dt0 = (float)(i * NB_SYMBOL_PERIOD) * f;             // ofdm_demodulator.cpp:675-676
        const bool do_corr = (i < NB_FRAME_SYMBOLS) && (i < out1 || i == NB_FRAME_SYMBOLS - 1);
        // ---- PLL ----
        PRIO_AT(0);
        const f4 w1_1 = w1p[0], w1_2 = w1p[1], w1_3 = w1p[2];
        f2 a[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float base = dt0 + gf[k];                              // apply_pll.cpp:103
            a[k]     = pll1(mk2(v[k].x, v[k].y), base, step0);
            a[4 + k] = pll1(mk2(v[k].z, v[k].w), base, step1);
        }

        // ---- cyclic prefix correlation: tail (slot k=3) x conj(head), fixed 256-leaf tree ----
        PRIO_AT(1);
        if (do_corr) {                                                   // uniform per workgroup
            f2 p = mk2(0.0f, 0.0f);
            if (t >= 4) {
                const float base = dt0 + gf[4];
                const f2 h0 = pll1(mk2(h.x, h.y), base, step0);
                const f2 h1 = pll1(mk2(h.z, h.w), base, step1);
                p = conj_mul(a[3], h0) + conj_mul(a[7], h1);
            }
            const float ps = wave_tree_sum_pair(p.x, p.y);   // lane 0: sum of p.x, lane 32: sum of p.y
            // (two sets: the next symbol's sums may be written before thread 0 has read these)
            if ((lane & 31) == 0) reinterpret_cast<float*>(red + 4 * (i & 1) + wave)[lane >> 5] = ps;
        }
        // ---- pass 1: radix 4 on positions p + 512 j (p = 2t, 2t+1); outputs stay in place ----
        {
            f2 b0, b1, b2, b3, c0, c1, c2, c3;
            dft4(a[0], a[1], a[2], a[3], b0, b1, b2, b3);
            dft4(a[4], a[5], a[6], a[7], c0, c1, c2, c3);
            b1 = cmul(b1, mk2(w1_1.x, w1_1.y)); b2 = cmul(b2, mk2(w1_2.x, w1_2.y)); b3 = cmul(b3, mk2(w1_3.x, w1_3.y));
            c1 = cmul(c1, mk2(w1_1.z, w1_1.w)); c2 = cmul(c2, mk2(w1_2.z, w1_2.w)); c3 = cmul(c3, mk2(w1_3.z, w1_3.w));
            // (the empty statement orders the arithmetic above against the barrier below: the compiler is otherwise free to sink
            // the PLL behind it, which puts the skew wait back in front of the longest arithmetic phase)
            asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b1.x), "+v"(b1.y), "+v"(b2.x), "+v"(b2.y), "+v"(b3.x), "+v"(b3.y),
                              "+v"(c0.x), "+v"(c0.y), "+v"(c1.x), "+v"(c1.y), "+v"(c2.x), "+v"(c2.y), "+v"(c3.x), "+v"(c3.y));
            // every wave is past its use of bufA / its transpose patch / obuf for the previous symbol.  This barrier sits HERE and not
            // at the end of the symbol: the PLL and the radix-4 arithmetic above need no LDS, so a wave that finished the previous
            // symbol early runs them while the slower waves catch up -- one skew-absorbing barrier per symbol instead of two
            __syncthreads();
            PRIO_AT(2);
            // the previous symbol's soft bits leave from here, not from the end of that symbol: a store issued after the prefetch
            // below would have to complete before the prefetched samples count as arrived (in-order vmcnt).  obuf is complete since
            // the barrier above and is not written again before the barrier below; its read goes ahead of the exchange writes so
            // that the HBM store does not wait for them
            const bool row_due = (i - 1 > out0 && i - 1 < NB_FRAME_SYMBOLS);
            u4v row = u4v{0u, 0u, 0u, 0u};
            if (row_due) row = row_read();
            f4* dst = reinterpret_cast<f4*>(bufA + 2 * t);
            dst[0]                  = f4{b0.x, b0.y, c0.x, c0.y};        // block j of 512 positions starts at j * WAVE_PATCH
            dst[WAVE_PATCH / 2]     = f4{b1.x, b1.y, c1.x, c1.y};
            dst[WAVE_PATCH]         = f4{b2.x, b2.y, c2.x, c2.y};
            dst[3 * WAVE_PATCH / 2] = f4{b3.x, b3.y, c3.x, c3.y};
            if (row_due) row_write(i - 2, row);
        }
        __syncthreads();                       // the only cross-wave exchange of the transform
        PRIO_AT(3);
        if (do_corr && t == 0) {
            const f2* rr = red + 4 * (i & 1);
            const f2 r0 = rr[0], r1 = rr[1], r2 = rr[2], r3 = rr[3];
            const f2 rsum = (r0 + r1) + (r2 + r3);
            cp_corr[(size_t)frame * NB_FRAME_SYMBOLS + i] = rsum;
            if (phase_tail) ph_corr[i] = rsum;
        }
        // this symbol's samples are consumed: the next symbol's loads go into the same registers now
        if (i < sym_end) load_symbol(i + 1, v, h);

        // ---- pass 2: radix 8 inside this wave's 512-point block ----
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = bufA[rd2 + 64 * j];
        dft8(a);
        patch[ta_w] = a[0];
#pragma unroll
        for (int k = 1; k < 8; k++) patch[ta_w + 72 * k] = cmul(a[k], w2[k - 1]);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = patch[ta_r + 8 * j];
        wave_lds_fence();

        // ---- pass 3: radix 8 inside 64-point blocks ----
        dft8(a);
        patch[tb_w] = a[0];
#pragma unroll
        for (int k = 1; k < 8; k++) patch[tb_w + 9 * k] = cmul(a[k], W3_LDS ? w3p[8 * (k - 1)] : w3[k - 1]);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = patch[tb_r + j];
        wave_lds_fence();

        // ---- pass 4: radix 8; thread ends with bins Kb + 256 k ----
        dft8(a);
        PRIO_AT(4);

        if (fft_out != nullptr) {
            f2* dst = fft_out + ((size_t)frame * (NB_FRAME_SYMBOLS + 1) + i) * NB_FFT + Kb;
#pragma unroll
            for (int k = 0; k < 8; k++) dst[256 * k] = a[k];
        }

        cur[0] = (Kb == 0) ? a[3] : a[0];
        cur[1] = a[1]; cur[2] = a[2]; cur[3] = a[5]; cur[4] = a[6]; cur[5] = a[7];

        const bool emit = (i > out0) && (i < NB_FRAME_SYMBOLS);
        if (emit) {
            // ---- DQPSK (X_{i-1} * conj(X_i)) + soft bits, scattered to their de-interleaved positions ----
            f2 dq[6];
            float An[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const f2 d = conj_mul(pv[k], cur[k]);
                if (dqpsk_out != nullptr) {                              // GetFrameDataVec() view, natural carrier order
                    const int cbase[6] = {767, 1023, 1279, 0, 256, 512};
                    const int c = (k == 0 && Kb == 0) ? 1535 : (cbase[k] + Kb);
                    dqpsk_out[((size_t)frame * (NB_FRAME_SYMBOLS - 1) + (i - 1)) * 1536 + c] = d;
                }
                const float ar = __builtin_fabsf(d.x), ai = __builtin_fabsf(d.y);
                dq[k] = d;
                An[k] = (ar < ai) ? ai : ar;                             // std::max, ofdm_demodulator.cpp:882
            }
            // Two IEEE divisions per carrier are 1/7 of the kernel's instructions.  For A in [2^-60, 2^60] and |n| <= A the
            // compiler's expansion of n / A (v_div_scale x2, v_rcp, 2 + 4 fma, v_div_fmas, v_div_fixup) never scales where it
            // matters: the scaled cases (quotient below 2^-43) end as soft bit 0 either way.  The reciprocal refinement then
            // depends on A only and is shared by both quotients; outside the range (or NaN) the plain expressions run.
            const float amin = __builtin_fminf(__builtin_fminf(__builtin_fminf(An[0], An[1]), An[2]), __builtin_fminf(__builtin_fminf(An[3], An[4]), An[5]));
            const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(An[0], An[1]), An[2]), __builtin_fmaxf(__builtin_fmaxf(An[3], An[4]), An[5]));
            // natural order: every byte goes to LDS as soon as it exists; class order keeps the twelve until the (uniform) choice of
            // the position set, so that the choice is one branch and not twelve selects
            int bx[CLASSED ? 6 : 1], by[CLASSED ? 6 : 1];
            auto put = [&](const int k, const int vx, const int vy) __attribute__((always_inline)) {
                if constexpr (CLASSED) { bx[k] = vx; by[k] = vy; }
                else { obuf[pos[k]] = (int8_t)vx; obuf[pos[k] + 1536] = (int8_t)vy; }
            };
            if (amin >= 0x1p-60f && amax <= 0x1p60f) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const float A = An[k], nx = dq[k].x, ny = -dq[k].y;
                    float r = __builtin_amdgcn_rcpf(A);
                    r = __builtin_fmaf(__builtin_fmaf(-A, r, 1.0f), r, r);
                    float qx = nx * r, qy = ny * r;
                    qx = __builtin_fmaf(__builtin_fmaf(-A, qx, nx), r, qx);
                    qy = __builtin_fmaf(__builtin_fmaf(-A, qy, ny), r, qy);
                    qx = __builtin_fmaf(__builtin_fmaf(-A, qx, nx), r, qx);
                    qy = __builtin_fmaf(__builtin_fmaf(-A, qy, ny), r, qy);
                    put(k, to_vbit(qx), to_vbit(qy));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) put(k, to_vbit(dq[k].x / An[k]), to_vbit(-(dq[k].y / An[k])));
            }
            if constexpr (CLASSED) {
                if (i - 1 >= NB_FIC_SYMBOLS) {                                    // (uniform) MSC symbol in class order
#pragma unroll
                    for (int k = 0; k < 6; k++) { obuf[posc[k]] = (int8_t)bx[k]; obuf[posc[k] + 96] = (int8_t)by[k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 6; k++) { obuf[pos[k]] = (int8_t)bx[k]; obuf[pos[k] + 1536] = (int8_t)by[k]; }
                }
            }
        }
        if (emit && i == sym_end) { __syncthreads(); store_row(i - 1); }  // (otherwise stored by the next symbol, ahead of its prefetch)
        PRIO_AT(5);
    };
    // two symbols per trip so that the bins kept for the next DQPSK change hands by name, not by 12 register moves
    for (int i = out0; i <= sym_end; i += 2) {
        symbol(i, prev, keep);
        if (i + 1 <= sym_end) symbol(i + 1, keep, prev);
    }
    if (phase_tail) {
        float* ang = reinterpret_cast<float*>(ph_corr + 80);
        __syncthreads();
        if (t < NB_FRAME_SYMBOLS) { const f2 c = ph_corr[t]; ang[t] = atan2_det(c.y, c.x); }
        __syncthreads();
        if (t == 0) {
            float total = 0.0f;
            for (int k = 0; k < NB_FRAME_SYMBOLS; k++) total += ang[k];                 // the reference's sequential sum (:606-618)
            if (pt.total_phase) pt.total_phase[frame] = total;
            if (pt.fine_freq) { float* ff = pt.fine_freq + (size_t)frame * pt.fine_stride; *ff = fine_freq_update(*ff, total, pt.beta, NB_FRAME_SYMBOLS, NB_FFT); }
        }
    }
}

// one thread per frame: sequential sum of the 76 per-symbol phase errors, then the fine-frequency IIR
// (ofdm_demodulator.cpp:606-618, :779-824, :829-840)
// one wave per frame: the per-symbol atan2 in parallel, then lane 0 adds them in symbol order (the reference's sequential
// sum, ofdm_demodulator.cpp:606-618) and runs the fine-frequency IIR (:779-824, :829-840)
__global__ __launch_bounds__(64)
void ofdm_phase_kernel(const f2* __restrict__ cp_corr, int n_frames, float beta,
                       float* __restrict__ total_phase, float* __restrict__ fine_freq, int fine_freq_stride,
                       const dabgpu_frame_desc* __restrict__ desc, int n_sym, int n_fft, const dabgpu_sync_state* __restrict__ sync)
{
    __shared__ float ph[160];                       // n_sym <= 153
    const int fr = blockIdx.x, lane = threadIdx.x;
    if (fr >= n_frames) return;
    if (desc != nullptr && desc[fr].slot < 0) return;
    if (sync != nullptr && !sync[fr].sync_valid) return;       // dabgpu_ofdm_sync_demod_frames: no frame was demodulated
    const f2* c = cp_corr + (size_t)fr * n_sym;
    for (int i = lane; i < n_sym; i += 64) { const f2 v = c[i]; ph[i] = atan2_det(v.y, v.x); }
    __syncthreads();
    if (lane != 0) return;
    float total = 0.0f;
    for (int i = 0; i < n_sym; i++) total += ph[i];
    if (total_phase) total_phase[fr] = total;
    if (fine_freq) fine_freq[(size_t)fr * fine_freq_stride] = fine_freq_update(fine_freq[(size_t)fr * fine_freq_stride], total, beta, n_sym, n_fft);
}

}  // namespace dabgpu

// ---- launchers (called from dabgpu_abi.hip) ----
// src: 0 interleaved complex float, 1 raw_u8 / wav pcm8, 2 raw_s8, 3 raw_s16l / wav pcm16
extern "C" hipError_t dabgpu_launch_ofdm_demod(const void* d_iq, int src, const float* d_freq, int8_t* d_bits, float* d_cp_corr,
                                               float* d_fft, float* d_dqpsk, const float* d_tw, const uint16_t* d_inv_map,
                                               int n_frames, int sym_per_chunk, size_t bits_frame_stride,
                                               const dabgpu_frame_desc* d_desc, const void* d_tail, size_t tail_stride,
                                               int classed, hipStream_t stream, float* d_total_phase, float* d_fine_freq, float beta,
                                               const void* d_prev_tail, size_t frame_stride_samples, dabgpu_sync_state* d_sync, int prs_offset)
{
    using namespace dabgpu;
    if (classed && (d_fft != nullptr || d_dqpsk != nullptr)) return hipErrorInvalidValue;   // soft bits only
    if (bits_frame_stride == 0) bits_frame_stride = NB_FRAME_BITS;
    // default: three runs per frame (one extra FFT per run).  A whole frame per workgroup (75) is 1 % faster when 1024 frames are
    // exactly one round of a 256-CU chip (and lets the phase tail run inside the kernel) but 25 % slower on boxes whose CUs do not
    // all run at one speed (workgroup lifetimes 0.33 .. 0.53 ms in one static round: 0.536 against 0.427 ms): callers that care
    // time both (bench.py does) and pass it
    if (sym_per_chunk <= 0 || sym_per_chunk > 75) sym_per_chunk = 25;
    const int chunks = (75 + sym_per_chunk - 1) / sym_per_chunk;
    const size_t lds = DEMOD_LDS_BYTES;
    const bool views_ = (d_fft != nullptr) || (d_dqpsk != nullptr);
    // the phase tail runs inside the kernel when one workgroup walks the whole frame; otherwise as its own launch below
    // frames one after the other unless the caller gives a stride; with sync records the fine-frequency words are theirs
    if (d_desc != nullptr && (d_sync != nullptr || frame_stride_samples != 0)) return hipErrorInvalidValue;
    const demod_frame_src fs = {frame_stride_samples ? frame_stride_samples : (size_t)NB_FRAME_SAMPLES, d_sync, prs_offset};
    int fine_stride = 1;
    if (d_sync != nullptr) { d_fine_freq = &d_sync->freq_fine; fine_stride = (int)(sizeof(dabgpu_sync_state) / sizeof(float)); }
    const bool fuse = (d_total_phase != nullptr || d_fine_freq != nullptr) && chunks == 1 && d_desc == nullptr && !views_;
    const demod_phase_tail pt = {fuse ? d_total_phase : nullptr, fuse ? d_fine_freq : nullptr, beta, fine_stride};
    const dim3 grid((unsigned)(n_frames * chunks));
#define DABGPU_LAUNCH_V(SRC, BANK, VIEWS) hipLaunchKernelGGL((ofdm_demod_kernel<SRC, BANK, VIEWS>), grid, dim3(256), lds, stream, \
                       d_iq, d_freq, d_bits, reinterpret_cast<f2*>(d_cp_corr), \
                       reinterpret_cast<f2*>(d_fft), reinterpret_cast<f2*>(d_dqpsk), reinterpret_cast<const f2*>(d_tw), d_inv_map, \
                       n_frames, sym_per_chunk, chunks, bits_frame_stride, d_desc, d_tail, tail_stride, pt, d_prev_tail, fs)
    const bool views = (d_fft != nullptr) || (d_dqpsk != nullptr);
#define DABGPU_LAUNCH(SRC, BANK) do { if (views) DABGPU_LAUNCH_V(SRC, BANK, true); else DABGPU_LAUNCH_V(SRC, BANK, false); } while (0)
#define DABGPU_LAUNCH_CB(SRC, BANK) hipLaunchKernelGGL((ofdm_demod_kernel<SRC, BANK, false, true>), grid, dim3(256), lds, stream, \
                       d_iq, d_freq, d_bits, reinterpret_cast<f2*>(d_cp_corr), static_cast<f2*>(nullptr), static_cast<f2*>(nullptr), \
                       reinterpret_cast<const f2*>(d_tw), d_inv_map, n_frames, sym_per_chunk, chunks, bits_frame_stride, d_desc, d_tail, tail_stride, pt, d_prev_tail, fs)
#define DABGPU_LAUNCH_C(SRC) do { if (d_desc != nullptr) DABGPU_LAUNCH_CB(SRC, true); else DABGPU_LAUNCH_CB(SRC, false); } while (0)
    switch (src) {
    case SRC_C32: if (classed) DABGPU_LAUNCH_C(SRC_C32); else if (d_desc != nullptr) DABGPU_LAUNCH(SRC_C32, true); else DABGPU_LAUNCH(SRC_C32, false); break;
    case SRC_U8: if (classed) DABGPU_LAUNCH_C(SRC_U8); else if (d_desc != nullptr) DABGPU_LAUNCH(SRC_U8, true); else DABGPU_LAUNCH(SRC_U8, false); break;
    case SRC_S8: if (classed) DABGPU_LAUNCH_C(SRC_S8); else if (d_desc != nullptr) DABGPU_LAUNCH(SRC_S8, true); else DABGPU_LAUNCH(SRC_S8, false); break;
    case SRC_S16: if (classed) DABGPU_LAUNCH_C(SRC_S16); else if (d_desc != nullptr) DABGPU_LAUNCH(SRC_S16, true); else DABGPU_LAUNCH(SRC_S16, false); break;
    default: return hipErrorInvalidValue;
    }
#undef DABGPU_LAUNCH_C
#undef DABGPU_LAUNCH_CB
#undef DABGPU_LAUNCH
#undef DABGPU_LAUNCH_V
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !fuse && (d_total_phase != nullptr || d_fine_freq != nullptr) && d_desc == nullptr) {
        hipLaunchKernelGGL(ofdm_phase_kernel, dim3((unsigned)n_frames), dim3(64), 0, stream, reinterpret_cast<const f2*>(d_cp_corr), n_frames, beta,
                           d_total_phase, d_fine_freq, fine_stride, static_cast<const dabgpu_frame_desc*>(nullptr), NB_FRAME_SYMBOLS, NB_FFT,
                           static_cast<const dabgpu_sync_state*>(d_sync));
        e = hipGetLastError();
    }
    return e;
}

extern "C" hipError_t dabgpu_launch_ofdm_phase(const float* d_cp_corr, int n_frames, float beta, float* d_total_phase,
                                               float* d_fine_freq, int fine_freq_stride, const dabgpu_frame_desc* d_desc, int n_sym, int n_fft,
                                               hipStream_t stream)
{
    using namespace dabgpu;
    const dim3 grid((unsigned)n_frames);
    hipLaunchKernelGGL(ofdm_phase_kernel, grid, dim3(64), 0, stream,
                       reinterpret_cast<const f2*>(d_cp_corr), n_frames, beta, d_total_phase, d_fine_freq, fine_freq_stride, d_desc, n_sym, n_fft,
                       static_cast<const dabgpu_sync_state*>(nullptr));
    return hipGetLastError();
}
