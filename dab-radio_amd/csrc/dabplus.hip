// dabplus.hip -- the DAB+ outer code on the device: what AAC_Frame_Processor does between the channel
// decoder's bytes and the AAC access units (src/dab/audio/aac_frame_processor.cpp:127-361):
//   super-frame acquisition on the fire code (:162-166, :178-189), accumulation of 5 logical frames (:191-197),
//   RS(120,110) decoding of the byte-interleaved columns (:323-361; decoder: src/dab/algorithms/reed_solomon_decoder.cpp,
//   Berlekamp-Massey / Chien / Forney over GF(2^8), p(x) = x^8+x^4+x^3+x^2+1, roots alpha^0..alpha^9, shortened by 135),
//   fire code of the corrected super frame (:206-209), header walk (:215-283) and access-unit CRCs (:286-317),
//   re-acquisition after 10 failed super frames (:151-154).
// One wavefront per stream (an (ensemble, sub-channel) pair); its acquisition state and the super frame being collected
// live in HBM between calls.  Syndromes: every lane adds the terms of a share of one codeword's symbols for all ten roots
// (XOR shuffles fold the shares); a codeword with a non-zero syndrome gets its error locator from ONE lane (Berlekamp-Massey,
// all codewords of the super frame in parallel), the Chien search and Forney's corrections are dealt over all 64 lanes; the
// access-unit CRCs are shared by 8-32 lanes each.  LDS table look-ups throughout (GF(2^8) log / exp, CRC-16).
// Integer / byte work, bit-exact by construction; checked against the oracle and the reference-generated vectors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include <vector>

#include "dabgpu.h"
#include "dabgpu_internal.h"

namespace dabgpu {

constexpr int RS_N = 120, RS_ROOTS = 10, RS_PAD = 135;
constexpr int DP_MAX_FRAME_BYTES = 1536;                 // 5 n / 120 <= 64 codewords = one per lane
constexpr int DP_MAX_SF = 5 * DP_MAX_FRAME_BYTES;

struct GfTables { uint8_t exp[512]; uint8_t log[256]; };
constexpr GfTables make_gf() {
    GfTables t{};
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        t.exp[i] = (uint8_t)x; t.exp[i + 255] = (uint8_t)x; t.log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100u) x ^= 0x11Du;
    }
    t.exp[510] = t.exp[0]; t.exp[511] = t.exp[1];
    t.log[0] = 0;
    return t;
}
__constant__ GfTables GF_TABLES = make_gf();

// fire code x^16+x^14+x^13+x^12+x^11+x^5+x^3+x^2+x+1 over the 72 bits after the check word, zero start value: message bit b (0 = MSB of
// the first byte) contributes x^(16 + 71 - b) mod p(x); the check word is the XOR of the contributions of the set bits
struct FireTab { uint16_t w[72]; };
constexpr FireTab make_fire_tab() {
    FireTab t{};
    unsigned c = 0x782Fu;                                    // x^16 mod p
    for (int p = 0; p < 72; p++) {
        t.w[71 - p] = (uint16_t)c;
        c = (c & 0x8000u) ? (((c << 1) ^ 0x782Fu) & 0xFFFFu) : ((c << 1) & 0xFFFFu);
    }
    return t;
}
__constant__ FireTab FIRE_TAB = make_fire_tab();

struct DpLds {
    uint8_t exp[512];
    uint8_t log[256];
    // CRC-16 x^16+x^12+x^5+1 table, MSB first (access units) -- or, while a super frame's codewords are being corrected, one record per
    // codeword: error-locator coefficients C_1..C_10, then (word at +12) degree | roots found << 8.  The table is rebuilt afterwards.
    union {
        uint16_t crc_tab[256];
        uint8_t rs_rec[64 * 16];
    };
    uint8_t sf[DP_MAX_SF];
    uint8_t syn[64 * RS_ROOTS];
    int flags[4];
    int au[8];                       // access-unit start offsets of the super frame being checked
};

struct DpState {                     // per stream, persistent
    int wait_frame_start, curr_dab_frame, prev_n, synced, desync_count;
};

__device__ __forceinline__ uint8_t gmul(const DpLds& L, uint8_t a, uint8_t b) { return (a && b) ? L.exp[L.log[a] + L.log[b]] : (uint8_t)0; }
__device__ __forceinline__ uint8_t gdiv(const DpLds& L, uint8_t a, uint8_t b) { return a ? L.exp[L.log[a] + 255 - L.log[b]] : (uint8_t)0; }
__device__ __forceinline__ uint8_t gpow(const DpLds& L, int e) { e %= 255; if (e < 0) e += 255; return L.exp[e]; }

// RS(120,110) decoding of a super frame's codewords (restates dab_rs120_decode of the oracle), in two phases:
// rs_locator: ONE lane per codeword with a non-zero syndrome runs Berlekamp-Massey and leaves the error locator (C_1..C_10, degree) in the
// codeword's record and the evaluator omega = S C mod x^deg in place of the syndromes;
// rs_search_and_correct: the Chien search over alpha^1..alpha^255 and Forney's formula are shared by ALL lanes -- the wavefront's lanes are
// dealt to the codewords (64 / n_rs each), every lane scans its share of the 255 candidates and corrects the symbol of each root it
// finds, counting it in the record.  A locator of degree d has at most d roots, so scanning all candidates finds what the reference's scan
// (which stops after d) finds; corrections made for a codeword that then turns out uncorrectable (roots != degree) are undone by the
// caller from the received bytes.
__device__ void rs_locator(DpLds& L, int i) {
    uint8_t S[RS_ROOTS];
#pragma unroll
    for (int r = 0; r < RS_ROOTS; r++) S[r] = L.syn[i * RS_ROOTS + r];
    uint8_t C[RS_ROOTS + 1], B[RS_ROOTS + 1], T[RS_ROOTS + 1];
#pragma unroll
    for (int k = 0; k <= RS_ROOTS; k++) { C[k] = (k == 0); B[k] = (k == 0); }
    int len = 0;
    for (int r = 1; r <= RS_ROOTS; r++) {
        uint8_t d = 0;
#pragma unroll
        for (int k = 0; k < RS_ROOTS; k++) if (k < r) d ^= gmul(L, C[k], S[(r - 1 - k) < 0 ? 0 : (r - 1 - k)]);
        if (d == 0) {
#pragma unroll
            for (int k = RS_ROOTS; k > 0; k--) B[k] = B[k - 1];
            B[0] = 0;
        } else {
            T[0] = C[0];
#pragma unroll
            for (int k = 0; k < RS_ROOTS; k++) T[k + 1] = (uint8_t)(C[k + 1] ^ gmul(L, d, B[k]));
            if (2 * len <= r - 1) {
                len = r - len;
#pragma unroll
                for (int k = 0; k <= RS_ROOTS; k++) B[k] = gdiv(L, C[k], d);
            } else {
#pragma unroll
                for (int k = RS_ROOTS; k > 0; k--) B[k] = B[k - 1];
                B[0] = 0;
            }
#pragma unroll
            for (int k = 0; k <= RS_ROOTS; k++) C[k] = T[k];
        }
    }
    int deg = 0;
#pragma unroll
    for (int k = 0; k <= RS_ROOTS; k++) if (C[k]) deg = k;
#pragma unroll
    for (int a = 0; a < RS_ROOTS; a++) {
        uint8_t t = 0;
#pragma unroll
        for (int b = 0; b < RS_ROOTS; b++) if (b <= a) t ^= gmul(L, S[a - b < 0 ? 0 : a - b], C[b]);
        L.syn[i * RS_ROOTS + a] = t;                                    // omega_a (the syndromes are done with)
    }
#pragma unroll
    for (int k = 1; k <= RS_ROOTS; k++) L.rs_rec[i * 16 + k - 1] = C[k];
    *reinterpret_cast<int*>(&L.rs_rec[i * 16 + 12]) = deg;
}

__device__ void rs_search_and_correct(DpLds& L, int i, int q, int per, int n_rs) {
    int* const rec = reinterpret_cast<int*>(&L.rs_rec[i * 16 + 12]);
    const int deg = *rec & 0xFF;
    if (deg == 0) return;
    const int span = (255 + per - 1) / per, x0 = 1 + q * span, x1 = (x0 + span - 1 < 255) ? (x0 + span - 1) : 255;
    uint8_t C[RS_ROOTS + 1];
    int reg[RS_ROOTS + 1];                                   // log C_k + x k (mod 255), advanced incrementally
    C[0] = 1;
#pragma unroll
    for (int k = 1; k <= RS_ROOTS; k++) {
        C[k] = L.rs_rec[i * 16 + k - 1];
        reg[k] = (L.log[C[k]] + k * (x0 - 1)) % 255;
    }
    for (int x = x0; x <= x1; x++) {
        uint8_t v = 1;
#pragma unroll
        for (int k = 1; k <= RS_ROOTS; k++) {
            if (k <= deg && C[k]) {
                reg[k] += k;
                if (reg[k] >= 255) reg[k] -= 255;
                v ^= L.exp[reg[k]];
            }
        }
        if (v) continue;
        atomicAdd(rec, 0x100);
        // Forney: e = omega(X^-1) X / C'(X^-1), first root alpha^0
        const int rk = x;
        uint8_t num1 = 0, den = 0;
#pragma unroll
        for (int a = RS_ROOTS - 1; a >= 0; a--) if (a < deg) num1 ^= gmul(L, L.syn[i * RS_ROOTS + a], gpow(L, a * rk));
        const uint8_t num2 = gpow(L, -rk);
        const int top = ((deg < RS_ROOTS - 1) ? deg : (RS_ROOTS - 1)) & ~1;
#pragma unroll
        for (int a = 8; a >= 0; a -= 2) if (a <= top) den ^= gmul(L, C[a + 1], gpow(L, a * rk));
        const int loc = rk - 1;
        if (num1 != 0 && loc >= RS_PAD) L.sf[i + (loc - RS_PAD) * n_rs] ^= gdiv(L, gmul(L, num1, num2), den);
    }
}

// fire code of the 9 bytes at x, by the whole wavefront: lane l takes message bits l and l + 64; every lane returns the check word
__device__ __forceinline__ uint16_t firecode_wave(const uint8_t* x, int lane) {
    uint32_t v = ((x[lane >> 3] >> (7 - (lane & 7))) & 1) ? FIRE_TAB.w[lane] : 0u;
    if (lane < 8 && ((x[8] >> (7 - lane)) & 1)) v ^= FIRE_TAB.w[64 + lane];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) v ^= (uint32_t)__shfl_xor((int)v, sft);
    return (uint16_t)v;
}

// a(x) b(x) mod x^16+x^12+x^5+1 (16-bit residues)
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        r ^= (0u - ((b >> i) & 1u)) & a;
        a = (a << 1) ^ ((0u - ((a >> 15) & 1u)) & 0x11021u);
    }
    return r;
}

// 9.6 KB of LDS per workgroup lets 16 single-wave workgroups share a CU = 4 per SIMD: hold the kernel to the 128 VGPRs that allows (the
// spilled values belong to the per-lane Berlekamp-Massey / Chien / Forney path; measured 0.224 / 0.169 / 0.153 ms per 18,432 clean super
// frames at 2 / 3 / 4 waves per SIMD, profiles/r04/ab_notes.md)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void dabplus_kernel(DpState* __restrict__ states, const uint8_t* __restrict__ frames, const unsigned long long* __restrict__ stream_offsets,
                    size_t frame_stride, const uint32_t* __restrict__ frame_bytes, int n_frames, uint8_t* __restrict__ sf_acc,
                    uint8_t* __restrict__ sf_out, size_t sf_out_stride, dabgpu_superframe_result* __restrict__ results, int max_sf,
                    int32_t* __restrict__ counts, int n_streams, const int32_t* __restrict__ active, int active_divisor)
{
    __shared__ DpLds L;
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= n_streams) return;
    if (active != nullptr && active[s / active_divisor] < 0) {           // nothing new for this stream's ensemble in this call
        if (lane == 0) { counts[4 * s] = 0; counts[4 * s + 1] = 0; counts[4 * s + 2] = 0; counts[4 * s + 3] = states[s].curr_dab_frame; }
        return;
    }
    for (int k = lane; k < 512; k += 64) L.exp[k] = GF_TABLES.exp[k];
    for (int k = lane; k < 256; k += 64) {
        L.log[k] = GF_TABLES.log[k];
        uint16_t c = (uint16_t)(k << 8);
#pragma unroll
        for (int j = 0; j < 8; j++) c = (c & 0x8000u) ? (uint16_t)((c << 1) ^ 0x1021u) : (uint16_t)(c << 1);
        L.crc_tab[k] = c;
    }
    DpState st = states[s];
    const int n = (int)frame_bytes[s];
    // what this kernel cannot hold is reported, not skipped in silence: logical frames above 1536 bytes (the reference accepts any
    // N >= 11, aac_frame_processor.cpp:129-137) -> counts[1] = -1; a super-frame slot smaller than 5 frames -> counts[1] = -2
    if (n > DP_MAX_FRAME_BYTES || (n >= 11 && (size_t)(5 * n) > sf_out_stride)) {
        if (lane == 0) { counts[4 * s] = 0; counts[4 * s + 1] = (n > DP_MAX_FRAME_BYTES) ? -1 : -2; counts[4 * s + 2] = 0; counts[4 * s + 3] = st.curr_dab_frame; }
        return;
    }
    uint8_t* acc = sf_acc + (size_t)s * DP_MAX_SF;
    const uint8_t* base = frames + stream_offsets[s];
    int n_sf = 0, n_wait_failed = 0, last_wait_crc = 0;
    __syncthreads();

    for (int f = 0; f < n_frames; f++) {
        const uint8_t* frame = base + (size_t)f * frame_stride;
        if (n < 11) continue;                                              // :129-137
        if (st.prev_n != n) { st.prev_n = n; st.curr_dab_frame = 0; st.wait_frame_start = 1; }       // :140-147
        if (st.desync_count >= 10) { st.desync_count = 0; st.synced = 0; }                           // :151-154
        if (st.synced) st.wait_frame_start = 0;                                                       // :158-160
        if (st.wait_frame_start) {                                                                    // :162-166
            const uint16_t rx = (uint16_t)((frame[0] << 8) | frame[1]);
            const uint16_t calc = firecode_wave(frame + 2, lane);
            const int ok = (rx == calc);
            if (!ok) last_wait_crc = (int)(((uint32_t)rx << 16) | calc);
            if (!ok) { n_wait_failed++; continue; }
            st.wait_frame_start = 0;
        }
        for (int k = lane; k < n; k += 64) acc[(size_t)st.curr_dab_frame * n + k] = frame[k];         // :191-197
        st.curr_dab_frame++;
        if (st.curr_dab_frame < 5) continue;

        // ---- ProcessSuperFrame (:199-321) ----
        st.wait_frame_start = 1;
        st.curr_dab_frame = 0;
        const int sf_bytes = 5 * n;
        const int n_rs = sf_bytes / RS_N;
        __syncthreads();                                                   // this wave's global writes of acc[] are visible to it
        for (int k = lane; k < sf_bytes; k += 64) L.sf[k] = acc[k];
        if (lane == 0) { L.flags[1] = -1; L.flags[2] = 0; }
        __syncthreads();
        // syndromes S_r(i) = sum_j c_j alpha^(r (119 - j)) -- what Horner's rule over the codeword (sy = sy alpha^r + c_j) evaluates.
        // When the codewords divide the wavefront (n_rs = 1, 2, 4, ... 64: every frame size of 24 x 2^k bytes, 192 among them) lane
        // (i, q) adds the terms of symbols j = q, q + 64 / n_rs, ... of codeword i for all ten roots -- independent table look-ups
        // instead of Horner's chain of 120 dependent ones -- and the partial sums of a codeword's lanes are folded by XOR shuffles
        // (GF(2^8) addition: exact, order-free).  Other sizes keep the (codeword, root) Horner form.
        if (n_rs > 0 && (64 % n_rs) == 0) {              // (frames below 24 bytes hold no codeword at all: n_rs = 0, nothing to correct)
            const int per = 64 / n_rs, i = lane % n_rs, q = lane / n_rs;
            uint32_t w0 = 0, w1 = 0, w2 = 0;                               // roots 0-3, 4-7, 8-9: one byte each
            for (int j = q; j < RS_N; j += per) {
                const uint8_t cj = L.sf[i + j * n_rs];
                if (cj) {
                    const int lc = L.log[cj], e = RS_N - 1 - j;
                    int t = 0;                                             // (r e) mod 255
#pragma unroll
                    for (int r = 0; r < RS_ROOTS; r++) {
                        const uint32_t term = L.exp[lc + t];
                        if (r < 4) w0 ^= term << (8 * r); else if (r < 8) w1 ^= term << (8 * (r - 4)); else w2 ^= term << (8 * (r - 8));
                        t += e; if (t >= 255) t -= 255;
                    }
                }
            }
            for (int sft = 32; sft >= n_rs; sft >>= 1) { w0 ^= (uint32_t)__shfl_xor((int)w0, sft); w1 ^= (uint32_t)__shfl_xor((int)w1, sft); w2 ^= (uint32_t)__shfl_xor((int)w2, sft); }
            if (q == 0) {
#pragma unroll
                for (int r = 0; r < RS_ROOTS; r++) L.syn[i * RS_ROOTS + r] = (uint8_t)(((r < 4) ? (w0 >> (8 * r)) : (r < 8) ? (w1 >> (8 * (r - 4))) : (w2 >> (8 * (r - 8)))) & 0xFFu);
            }
        } else {
            for (int w = lane; w < n_rs * RS_ROOTS; w += 64) {             // work item = (codeword i, root r)
                const int i = w / RS_ROOTS, r = w - i * RS_ROOTS;
                uint8_t sy = 0;
                for (int j = 0; j < RS_N; j++) {
                    const uint8_t m = sy ? L.exp[L.log[sy] + r] : (uint8_t)0;
                    sy = (uint8_t)(m ^ L.sf[i + j * n_rs]);
                }
                L.syn[w] = sy;
            }
        }
        __syncthreads();
        int any = 0;
        if (lane < n_rs) {
#pragma unroll
            for (int r = 0; r < RS_ROOTS; r++) any |= L.syn[lane * RS_ROOTS + r];
        }
        const bool decode = __any(any != 0);               // (clean super frames skip all of this and keep the CRC table)
        int my_cnt = 0;
        if (decode) {
            __syncthreads();                                // every lane has read its syndromes; the records take the table's place
            if (lane < n_rs) {
                if (any) rs_locator(L, lane);
                else *reinterpret_cast<int*>(&L.rs_rec[lane * 16 + 12]) = 0;
            }
            __syncthreads();
            const int per = 64 / n_rs;                      // lanes per codeword (1 <= n_rs <= 64 here: a non-zero syndrome exists)
            if (lane < per * n_rs) rs_search_and_correct(L, lane % n_rs, lane / n_rs, per, n_rs);
            __syncthreads();
            if (lane < n_rs) {
                const int w = *reinterpret_cast<const int*>(&L.rs_rec[lane * 16 + 12]), deg = w & 0xFF, found = w >> 8;
                my_cnt = (found == deg) ? deg : -1;
            }
            __syncthreads();
            for (int k = lane; k < 256; k += 64) {          // the CRC table back in its place
                uint16_t c = (uint16_t)(k << 8);
#pragma unroll
                for (int j = 0; j < 8; j++) c = (c & 0x8000u) ? (uint16_t)((c << 1) ^ 0x1021u) : (uint16_t)(c << 1);
                L.crc_tab[k] = c;
            }
        }
        __syncthreads();
        if (lane == 0) L.flags[3] = 0x7FFFFFFF;
        __syncthreads();
        if (my_cnt < 0) atomicMin(&L.flags[3], lane);
        if (my_cnt > 0) atomicAdd(&L.flags[2], my_cnt);
        __syncthreads();
        const int first_fail = L.flags[3];
        dabgpu_superframe_result res;
        memset(&res, 0, sizeof(res));
        res.rs_failed_index = (first_fail == 0x7FFFFFFF) ? -1 : first_fail;
        res.au_walk_stopped_at = -1;
        res.frame_index = f;
        bool good = (res.rs_failed_index < 0);
        if (!good) {
            // the reference stops at the first uncorrectable codeword (:336-341): later codewords keep their received
            // symbols and the corrected-symbol count only covers the codewords before it
            // (and the failed codeword itself: the shared search corrects as it finds roots, before their number is known)
            for (int k = lane; k < sf_bytes; k += 64) if ((k % n_rs) >= first_fail) L.sf[k] = acc[k];
            if (lane == 0) L.flags[2] = 0;
            __syncthreads();
            if (my_cnt > 0 && lane < first_fail) atomicAdd(&L.flags[2], my_cnt);
            __syncthreads();
        }
        res.rs_corrected = L.flags[2];
        if (good) {
            const uint16_t rx = (uint16_t)((L.sf[0] << 8) | L.sf[1]);
            const uint16_t calc = firecode_wave(L.sf + 2, lane);
            good = (rx == calc);
            res.firecode_ok = good ? 1 : 0;
            res.firecode_rx_calc = ((uint32_t)rx << 16) | calc;
        }
        if (!good) {
            st.desync_count++;
        } else {
            st.desync_count = 0;
            st.synced = 1;
            const int descriptor = L.sf[2];
            const int dac_rate = (descriptor >> 6) & 1, sbr = (descriptor >> 5) & 1;
            res.header_valid = 1;
            res.descriptor = descriptor;
            const int num_aus = dac_rate ? (sbr ? 3 : 6) : (sbr ? 2 : 4);
            res.num_aus = num_aus;
            int bit = 0;
#pragma unroll
            for (int a = 1; a < 6; a++) {
                if (a < num_aus) {
                    int v = 0;
                    for (int b = 0; b < 12; b++, bit++) v = (v << 1) | ((L.sf[3 + (bit >> 3)] >> (7 - (bit & 7))) & 1);
                    res.au_start[a] = v;
                }
            }
            res.au_start[0] = 3 + ((bit + 7) >> 3);
#pragma unroll
            for (int a = 2; a <= 6; a++) if (a == num_aus) res.au_start[a] = 110 * n_rs;
            // the walk stops at the first access unit that fails the bounds test (:291-297)
            int stop = -1;
#pragma unroll
            for (int a = 5; a >= 0; a--) {
                if (a < num_aus) {
                    const int nb_data = res.au_start[a + 1] - res.au_start[a] - 2;
                    if (nb_data < 0 || res.au_start[a + 1] >= sf_bytes) stop = a;
                }
            }
            res.au_walk_stopped_at = stop;
            const int walked = (stop < 0) ? num_aus : stop;
            // access-unit CRCs (x^16+x^12+x^5+1, start value 0xFFFF, inverted).  An access unit is shared by `per` lanes: with the
            // start value folded into its first two bytes the register is linear in the message, so the unit is padded at the FRONT
            // with zero bytes to per x c, lane q runs chunk q from a zero register (and x^(8c) beside it, on the same table), and the
            // chunks are joined pairwise -- left x^(8 c 2^k) + right -- in log2(per) rounds of shuffles.
            if (lane == 0) L.flags[0] = 0;
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a <= 6; a++) L.au[a] = res.au_start[a];
            }
            __syncthreads();
            {
                const int lg = (num_aus <= 2) ? 5 : (num_aus <= 4) ? 4 : 3, per = 1 << lg;
                const int a = lane >> lg, q = lane & (per - 1);
                const bool mine = a < walked;
                const int a0 = mine ? L.au[a] : 0, nb_data = mine ? (L.au[a + 1] - a0 - 2) : 0;
                uint32_t state = 0, m = 1;
                if (nb_data >= 2) {
                    const int c = (nb_data + per - 1) >> lg, z = per * c - nb_data;
                    for (int t = 0, k = q * c - z; t < c; t++, k++) {
                        uint32_t byte = (k >= 0) ? L.sf[a0 + k] : 0u;
                        if (k == 0 || k == 1) byte ^= 0xFFu;
                        state = ((state << 8) ^ L.crc_tab[((state >> 8) ^ byte) & 0xFFu]) & 0xFFFFu;
                        m = ((m << 8) ^ L.crc_tab[(m >> 8) & 0xFFu]) & 0xFFFFu;
                    }
                } else if (nb_data == 1 && q == per - 1) {
                    state = 0xFFFFu;
                    state = ((state << 8) ^ L.crc_tab[((state >> 8) ^ L.sf[a0]) & 0xFFu]) & 0xFFFFu;
                } else if (q == per - 1) state = 0xFFFFu;
                for (int d = 1; d < per; d <<= 1) {
                    const uint32_t left = (uint32_t)__shfl_xor((int)state, d);
                    if (q & d) state ^= crc_mulmod(left, m);
                    m = crc_mulmod(m, m);
                }
                if (mine && q == per - 1) {
                    const uint16_t crc = (uint16_t)(state ^ 0xFFFFu);
                    const uint16_t rx = (uint16_t)((L.sf[a0 + nb_data] << 8) | L.sf[a0 + nb_data + 1]);
                    if (rx == crc) atomicOr(&L.flags[0], 1 << a);
                    L.syn[2 * a] = (uint8_t)(crc >> 8); L.syn[2 * a + 1] = (uint8_t)(crc & 0xFF);       // syndromes are done with
                }
            }
            __syncthreads();
            res.au_crc_ok_mask = (uint32_t)L.flags[0];
#pragma unroll
            for (int a = 0; a < 6; a++) if (a < walked) res.au_crc_calc[a] = (uint16_t)((L.syn[2 * a] << 8) | L.syn[2 * a + 1]);
        }
        // hand the (corrected) super frame and its record to the caller
        if (n_sf < max_sf) {
            uint8_t* out = sf_out + ((size_t)s * max_sf + n_sf) * sf_out_stride;
            for (int k = lane; k < sf_bytes; k += 64) out[k] = L.sf[k];
            if (lane == 0) results[(size_t)s * max_sf + n_sf] = res;
        }
        n_sf++;
        __syncthreads();
    }
    if (lane == 0) {
        states[s] = st;
        counts[4 * s] = n_sf;
        counts[4 * s + 1] = n_wait_failed;
        counts[4 * s + 2] = last_wait_crc;
        counts[4 * s + 3] = st.curr_dab_frame;
    }
}

}  // namespace dabgpu

using namespace dabgpu;

struct dabgpu_dabplus_bank {
    dabgpu_ctx* ctx = nullptr;
    size_t n = 0;
    DpState* d_states = nullptr;
    uint8_t* d_acc = nullptr;
    // single-stream host-buffer path
    uint8_t* d_frame = nullptr; unsigned long long* d_off = nullptr; uint32_t* d_n = nullptr;
    uint8_t* d_sf = nullptr; dabgpu_superframe_result* d_res = nullptr; int32_t* d_counts = nullptr;   // d_counts [4]
    std::vector<void*> allocs;
};

extern "C" {

void dabgpu_dabplus_bank_destroy(dabgpu_dabplus_bank* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : b->allocs) (void)hipFree(p);
    delete b;
}

int dabgpu_dabplus_bank_reset(dabgpu_dabplus_bank* b, void* stream) {
    if (!b) { dabgpu_set_error("dabplus_bank_reset: null bank"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(b->ctx);
    // AAC_Frame_Processor's constructor state (:121-125): WAIT_FRAME_START, nothing collected, unsynchronised
    std::vector<DpState> init(b->n, DpState{1, 0, 0, 0, 0});
    int st = dabgpu_check_hip(hipMemcpyAsync(b->d_states, init.data(), b->n * sizeof(DpState), hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync");
    if (st) return st;
    return dabgpu_check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
}

int dabgpu_dabplus_bank_create(dabgpu_ctx* c, size_t n_streams, dabgpu_dabplus_bank** out) {
    if (!c || !out || n_streams == 0 || n_streams > (size_t)(1 << 22)) { dabgpu_set_error("dabplus_bank_create: invalid argument"); return DABGPU_ERR_INVALID_ARG; }
    DABGPU_BIND(c);
    dabgpu_dabplus_bank* b = new (std::nothrow) dabgpu_dabplus_bank();
    if (!b) return DABGPU_ERR_HIP;
    b->ctx = c; b->n = n_streams;
    int st = DABGPU_OK;
    auto alloc = [&](void** p, size_t bytes) {
        if (st) return;
        st = dabgpu_check_hip(hipMalloc(p, bytes), "hipMalloc(dabplus bank)");
        if (!st) b->allocs.push_back(*p);
    };
    alloc((void**)&b->d_states, n_streams * sizeof(DpState));
    alloc((void**)&b->d_acc, n_streams * (size_t)DP_MAX_SF);
    alloc((void**)&b->d_frame, DP_MAX_FRAME_BYTES);
    alloc((void**)&b->d_off, sizeof(unsigned long long));
    alloc((void**)&b->d_n, sizeof(uint32_t));
    alloc((void**)&b->d_sf, DP_MAX_SF);
    alloc((void**)&b->d_res, sizeof(dabgpu_superframe_result));
    alloc((void**)&b->d_counts, 4 * sizeof(int32_t));
    if (!st) st = dabgpu_check_hip(hipMemsetAsync(b->d_off, 0, sizeof(unsigned long long), c->stream), "hipMemsetAsync");
    if (!st) st = dabgpu_dabplus_bank_reset(b, c->stream);
    if (st) { dabgpu_dabplus_bank_destroy(b); return st; }
    *out = b;
    return DABGPU_OK;
}

int dabgpu_dabplus_bank_process(dabgpu_dabplus_bank* b, const uint8_t* d_frames, const uint64_t* d_stream_offsets, size_t frame_stride_bytes,
                                const uint32_t* d_frame_bytes, int n_frames, uint8_t* d_superframes, size_t superframe_stride_bytes,
                                dabgpu_superframe_result* d_results, int max_superframes, int32_t* d_counts, void* stream) {
    return dabgpu_dabplus_bank_process_masked(b, d_frames, d_stream_offsets, frame_stride_bytes, d_frame_bytes, n_frames, d_superframes,
                                              superframe_stride_bytes, d_results, max_superframes, d_counts, nullptr, 1, stream);
}

int dabgpu_dabplus_bank_process_masked(dabgpu_dabplus_bank* b, const uint8_t* d_frames, const uint64_t* d_stream_offsets, size_t frame_stride_bytes,
                                       const uint32_t* d_frame_bytes, int n_frames, uint8_t* d_superframes, size_t superframe_stride_bytes,
                                       dabgpu_superframe_result* d_results, int max_superframes, int32_t* d_counts,
                                       const int32_t* d_active, int streams_per_flag, void* stream) {
    if (!b || !d_frames || !d_stream_offsets || !d_frame_bytes || !d_superframes || !d_results || !d_counts) {
        dabgpu_set_error("dabplus_bank_process: null argument"); return DABGPU_ERR_INVALID_ARG;
    }
    if (streams_per_flag < 1) { dabgpu_set_error("dabplus_bank_process_masked: streams_per_flag must be positive"); return DABGPU_ERR_INVALID_ARG; }
    if (n_frames <= 0) return DABGPU_OK;
    if (max_superframes < (n_frames + 4) / 5 || superframe_stride_bytes == 0) {
        dabgpu_set_error("dabplus_bank_process: max_superframes must be at least ceil(n_frames / 5)"); return DABGPU_ERR_INVALID_ARG;
    }
    DABGPU_BIND(b->ctx);
    hipLaunchKernelGGL(dabplus_kernel, dim3((unsigned)b->n), dim3(64), 0, (hipStream_t)stream, b->d_states, d_frames,
                       reinterpret_cast<const unsigned long long*>(d_stream_offsets), frame_stride_bytes, d_frame_bytes, n_frames, b->d_acc,
                       d_superframes, superframe_stride_bytes, d_results, max_superframes, d_counts, (int)b->n, d_active, streams_per_flag);
    return dabgpu_check_hip(hipGetLastError(), "dabplus_kernel launch");
}

int dabgpu_dabplus_process_frame_host_sync(dabgpu_dabplus_bank* b, const uint8_t* h_frame, uint32_t n_bytes, int* superframe_done,
                                           int* firecode_wait_failed, uint32_t* firecode_wait_rx_calc, dabgpu_superframe_result* h_result,
                                           uint8_t* h_superframe) {
    if (!b || !h_frame || !superframe_done || !firecode_wait_failed || !h_result || !h_superframe) {
        dabgpu_set_error("dabplus_process_frame_host_sync: null argument"); return DABGPU_ERR_INVALID_ARG;
    }
    if (b->n != 1) { dabgpu_set_error("dabplus_process_frame_host_sync: needs a bank of one stream"); return DABGPU_ERR_INVALID_ARG; }
    *superframe_done = 0; *firecode_wait_failed = 0;
    if (n_bytes > (uint32_t)DP_MAX_FRAME_BYTES) {
        dabgpu_set_error("dabplus_process_frame_host_sync: logical frames of %u bytes are not supported (<= %d)", n_bytes, DP_MAX_FRAME_BYTES);
        return DABGPU_ERR_UNSUPPORTED;
    }
    dabgpu_ctx* c = b->ctx;
    DABGPU_BIND(c);
    DABGPU_HOST_LOCK(b->ctx);
    hipStream_t s = c->stream;
    int st;
#define CK(call) do { st = dabgpu_check_hip((call), #call); if (st) return st; } while (0)
    CK(hipMemcpyAsync(b->d_frame, h_frame, n_bytes, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(b->d_n, &n_bytes, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if ((st = dabgpu_dabplus_bank_process(b, b->d_frame, reinterpret_cast<const uint64_t*>(b->d_off), 0, b->d_n, 1, b->d_sf, DP_MAX_SF, b->d_res, 1,
                                          b->d_counts, s))) return st;
    int32_t counts[4];
    CK(hipMemcpyAsync(counts, b->d_counts, sizeof(counts), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    *superframe_done = counts[0];
    *firecode_wait_failed = counts[1];
    if (firecode_wait_rx_calc) *firecode_wait_rx_calc = (uint32_t)counts[2];
    if (counts[0]) {
        CK(hipMemcpyAsync(h_result, b->d_res, sizeof(dabgpu_superframe_result), hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(h_superframe, b->d_sf, 5 * (size_t)n_bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
    }
#undef CK
    return DABGPU_OK;
}

}  // extern "C"
