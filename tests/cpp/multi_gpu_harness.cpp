// multi_gpu_harness.cpp -- the C++ multi-device host of the batch path: one std::thread per listed device, each with its own
// dabgpu contexts, streams and block of ensembles, running BASELINE configs[4]'s step (PRS synchronisation -> OFDM demodulation at the
// position and with the carrier offset it found, into the frame-history ring -> fine-frequency update -> FIC Viterbi -> MSC time
// de-interleave + Viterbi + descramble for 18 x 48 CU EEP 3-A) with two transmission frames in flight.  Every multiplex has its own
// carrier offset (+-5 kHz) and timing offset (+-100 samples); `--aligned` = frame-aligned input, synchronisation bypassed (rounds 1-3).
// Ensembles are independent, so there is no collective: this is the one-process form of what `bench.py --gpus N` does with one
// process per GPU (replaces the per-frame fan-out of src/basic_radio/basic_radio.cpp:51-62 across a node).
//
//   multi_gpu_harness --devices 0,1,2,3 [--ensembles 8192] [--steps 10] [--distinct 8] [--inflight 2] [--identical] [--aligned]
// `--devices 0,0` runs two workers on ONE GPU (the -m gpu test: tests/test_gpu_multi_device.py).  Every worker generates its
// multiplexes itself (seeded: energy dispersal, convolutional code, puncturing, FIB CRCs, QPSK / differential modulation / IFFT /
// cyclic prefix -- the transmit side, written independently of the path under test), checks every decoded FIB and sub-channel byte of
// its last frames against what it transmitted and prints a digest of them.  One JSON line: frames/s = ensembles x steps x workers /
// the slowest worker's time (all workers start together).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dabgpu.h"

namespace {

constexpr int N_SUB = 18, SUB_BYTES = 192, SUB_CU = 48, FRAME_BITS = 230400, FRAME_SAMPLES = 196608;
// synchronised path: a receiver's slice = SLICE_LEAD samples before the expected PRS position, then room for the latest frame start the
// synchroniser can report (dabgpu_ofdm_sync_demod_frames: stride >= offset + 1544 + 76 x 2552)
constexpr int SLICE_LEAD = 1024, SLICE_SAMPLES = SLICE_LEAD + 1544 + FRAME_SAMPLES, FRAME_BODY = 76 * 2552;
typedef std::complex<float> c32;

struct Rng {                                                   // xorshift64*
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint32_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (uint32_t)((s * 0x2545F4914F6CDD1Dull) >> 32); }
};

// energy dispersal (ETSI EN 300 401 clause 10): x^9 + x^5 + 1, all ones
std::vector<uint8_t> prbs_bytes(size_t n) {
    std::vector<uint8_t> out(n);
    unsigned reg = 0xFFFFu;
    for (size_t k = 0; k < n; k++) {
        unsigned b = 0;
        for (int i = 0; i < 8; i++) { const unsigned v = ((reg >> 8) ^ (reg >> 4)) & 1u; b |= v << (7 - i); reg = ((reg << 1) | v) & 0xFFFFu; }
        out[k] = (uint8_t)b;
    }
    return out;
}

uint16_t crc16(const uint8_t* p, size_t n) {                   // FIB CRC (clause 5.2.1): x^16 + x^12 + x^5 + 1, init and final inversion
    unsigned crc = 0xFFFFu;
    for (size_t i = 0; i < n; i++) {
        crc ^= (unsigned)p[i] << 8;
        for (int q = 0; q < 8; q++) crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
    }
    return (uint16_t)(crc ^ 0xFFFFu);
}

// mother code (clause 11.1): polynomials 133, 171, 145, 133 (octal), 6 tail bits; bytes MSB first
std::vector<uint8_t> conv_encode(const std::vector<uint8_t>& bytes) {
    static const int TAPS[4][5] = {{0, 2, 3, 5, 6}, {0, 1, 2, 3, 6}, {0, 1, 4, 6, -1}, {0, 2, 3, 5, 6}};
    const size_t n = bytes.size() * 8;
    std::vector<uint8_t> x(n + 12, 0), out(4 * (n + 6));
    for (size_t i = 0; i < n; i++) x[6 + i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
    for (size_t t = 0; t < n + 6; t++)
        for (int r = 0; r < 4; r++) {
            unsigned v = 0;
            for (int k = 0; k < 5; k++) if (TAPS[r][k] >= 0) v ^= x[6 + t - (size_t)TAPS[r][k]];
            out[4 * t + r] = (uint8_t)v;
        }
    return out;
}

// puncturing (clause 11.1.2): PI_n keeps 8 + n of every 32 mother bits; tail = the first 24 bits under PI_8's pattern
std::vector<uint8_t> puncture(const std::vector<uint8_t>& mother, const std::vector<std::pair<int, int>>& segments) {
    static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    std::vector<uint8_t> out;
    size_t m = 0;
    auto run = [&](int pi, int n_groups) {
        int cnt[8];
        for (int g = 0; g < 8; g++) cnt[g] = 1;
        for (int e = 0; e < pi; e++) cnt[order[e % 8]]++;
        for (int g = 0; g < n_groups; g++) for (int r = 0; r < cnt[g % 8]; r++) out.push_back(mother[m + 4 * (size_t)g + (size_t)r]);
        m += 4 * (size_t)n_groups;
    };
    for (const auto& s : segments) run(s.first, 32 * s.second);
    run(8, 6);
    return out;
}

void fft2048(std::vector<std::complex<double>>& a, bool inverse) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = 2.0 * M_PI / (double)len * (inverse ? 1.0 : -1.0);
        const std::complex<double> wl(std::cos(ang), std::sin(ang));
        for (size_t i = 0; i < n; i += len) {
            std::complex<double> w(1.0, 0.0);
            for (size_t k = 0; k < len / 2; k++) {
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v; a[i + k + len / 2] = u - v;
                w *= wl;
            }
        }
    }
}

struct Multiplex {
    std::vector<uint8_t> fibs;       // [4][96]
    std::vector<uint8_t> payload;    // [18][192]
    std::vector<c32> iq;             // one transmission frame, frame-buffer layout (76 symbols, then the NULL period at zero)
};

Multiplex make_multiplex(uint64_t seed, const float* prs, const int* mapper) {
    Multiplex M;
    Rng rng(seed);
    M.fibs.resize(4 * 96); M.payload.resize((size_t)N_SUB * SUB_BYTES);
    for (int g = 0; g < 4; g++)
        for (int f = 0; f < 3; f++) {
            uint8_t* fib = &M.fibs[(size_t)g * 96 + (size_t)f * 32];
            for (int i = 0; i < 30; i++) fib[i] = (uint8_t)rng.next();
            const uint16_t c = crc16(fib, 30);
            fib[30] = (uint8_t)(c >> 8); fib[31] = (uint8_t)c;
        }
    for (auto& b : M.payload) b = (uint8_t)rng.next();
    std::vector<uint8_t> bits((size_t)FRAME_BITS);
    const std::vector<uint8_t> pr = prbs_bytes(SUB_BYTES);
    for (int g = 0; g < 4; g++) {                                      // FIC: PI_16 x 21 blocks, PI_15 x 3 blocks, tail (clause 11.2)
        std::vector<uint8_t> d(96);
        for (int i = 0; i < 96; i++) d[(size_t)i] = M.fibs[(size_t)g * 96 + (size_t)i] ^ pr[(size_t)i];
        const std::vector<uint8_t> tx = puncture(conv_encode(d), {{16, 21}, {15, 3}});
        std::memcpy(&bits[(size_t)g * 2304], tx.data(), 2304);
    }
    for (int s = 0; s < N_SUB; s++) {                                  // EEP 3-A on 48 CU: PI_8 x 45 blocks, PI_7 x 3 blocks (clause 11.3.2)
        std::vector<uint8_t> d(SUB_BYTES);
        for (int i = 0; i < SUB_BYTES; i++) d[(size_t)i] = M.payload[(size_t)s * SUB_BYTES + (size_t)i] ^ pr[(size_t)i];
        const std::vector<uint8_t> tx = puncture(conv_encode(d), {{8, 45}, {7, 3}});
        for (int c = 0; c < 4; c++)                                    // every CIF repeats the logical frame: the time interleaver is in steady state
            std::memcpy(&bits[9216 + (size_t)c * 55296 + (size_t)s * 3072], tx.data(), 3072);
    }
    // OFDM (clauses 14.5-14.7): QPSK on de-interleaved positions, differential modulation from the phase reference symbol, IFFT, cyclic prefix
    M.iq.assign((size_t)FRAME_SAMPLES, c32(0.0f, 0.0f));
    std::vector<int> bin(1536);
    for (int n = 0; n < 1536; n++) bin[(size_t)n] = mapper[n] < 768 ? mapper[n] + (2048 - 768) : mapper[n] - 768 + 1;
    std::vector<std::complex<double>> cur(1536), spec(2048);
    for (int n = 0; n < 1536; n++) cur[(size_t)n] = std::complex<double>(prs[2 * bin[(size_t)n]], prs[2 * bin[(size_t)n] + 1]);
    const double a = 0.7071067811865476, scale = 1.0 / 39.2;
    for (int sym = 0; sym < 76; sym++) {
        for (auto& v : spec) v = 0.0;
        if (sym == 0) for (int k = 0; k < 2048; k++) spec[(size_t)k] = std::complex<double>(prs[2 * k], prs[2 * k + 1]);
        else {
            const uint8_t* b = &bits[(size_t)(sym - 1) * 3072];
            for (int n = 0; n < 1536; n++) {
                cur[(size_t)n] *= std::complex<double>((1.0 - 2.0 * b[n]) * a, (1.0 - 2.0 * b[1536 + n]) * a);
                spec[(size_t)bin[(size_t)n]] = cur[(size_t)n];
            }
        }
        fft2048(spec, true);                                           // unnormalised inverse = ifft x 2048
        c32* body = &M.iq[(size_t)sym * 2552];
        for (int k = 0; k < 2048; k++) body[504 + k] = c32((float)(spec[(size_t)k].real() * scale), (float)(spec[(size_t)k].imag() * scale));
        for (int k = 0; k < 504; k++) body[k] = body[2048 + k];
    }
    return M;
}

#define HIPCK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return false; } } while (0)
#define DABCK(call) do { const int s_ = (call); if (s_ != DABGPU_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #call, dabgpu_strerror(s_), dabgpu_last_error()); return false; } } while (0)

struct Barrier {
    std::mutex mu; std::condition_variable cv; int waiting = 0, n = 0, gen = 0;
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const int g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct Result {
    int device = 0; bool ok = false; double ms_per_step = 0.0;
    size_t free_at_probe = 0, free_at_end = 0;      // hipMemGetInfo of the worker's device after step --mem-probe-step and after the last step
    int spb = 0;
    long fib_mismatch = 0, msc_mismatch = 0, crc_pass = 0, crc_expected = 0;
    uint64_t digest = 0;
};

uint64_t fnv1a(uint64_t h, const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001B3ull; } return h; }

bool worker(int rank, int device, size_t E, int steps, int n_distinct, int inflight, int probe_step, bool synced, Barrier* bar, Result* R) {
    // rank < 0 (--identical): every worker carries the same ensembles (equal digests), else worker r starts at multiplex r
    if (rank < 0) rank = 0;
    R->device = device;
    HIPCK(hipSetDevice(device));
    std::vector<float> prs(2 * 2048);
    std::vector<int> mapper(1536);
    DABCK(dabgpu_get_prs_fft_ref(1, prs.data()));
    DABCK(dabgpu_get_carrier_mapper(1, mapper.data()));
    std::vector<Multiplex> mux;
    for (int d = 0; d < n_distinct; d++) mux.push_back(make_multiplex(1000u + (uint64_t)d, prs.data(), mapper.data()));   // the same on every worker

    const int H = inflight == 1 ? 5 : 8;
    const size_t stride = (size_t)H * FRAME_BITS, cif_out = (size_t)N_SUB * SUB_BYTES;
    std::vector<dabgpu_ctx*> ctx((size_t)inflight, nullptr);
    std::vector<hipStream_t> st((size_t)inflight, nullptr);
    for (int k = 0; k < inflight; k++) { DABCK(dabgpu_create(&ctx[(size_t)k], device, nullptr, nullptr)); HIPCK(hipStreamCreateWithFlags(&st[(size_t)k], hipStreamNonBlocking)); }
    // what a receiver is handed per transmission frame: the frame itself (aligned), or a slice around where it expects the frame, in
    // which the frame begins toff samples off and carries the multiplex's carrier offset
    const size_t in_samples = synced ? (size_t)SLICE_SAMPLES : (size_t)FRAME_SAMPLES;
    c32* d_iq = nullptr; int8_t* d_hist = nullptr; dabgpu_sync_state* d_states = nullptr;
    HIPCK(hipMalloc((void**)&d_iq, E * in_samples * sizeof(c32)));
    HIPCK(hipMalloc((void**)&d_hist, E * stride));
    HIPCK(hipMemset(d_hist, 0, E * stride));
    HIPCK(hipMalloc((void**)&d_states, E * sizeof(dabgpu_sync_state)));
    HIPCK(hipMemset(d_states, 0, E * sizeof(dabgpu_sync_state)));       // OFDM_Demod's constructed state: nothing found yet
    {
        c32* d_base = nullptr;
        HIPCK(hipMalloc((void**)&d_base, (size_t)n_distinct * in_samples * sizeof(c32)));
        std::vector<c32> slice;
        for (int d = 0; d < n_distinct; d++) {
            const c32* src = mux[(size_t)d].iq.data();
            if (synced) {
                Rng rng(777u + (uint64_t)d);
                const double cfo = ((double)rng.next() / 4294967296.0 * 2.0 - 1.0) * (5000.0 / 2.048e6);       // cycles per sample
                const int toff = (int)(rng.next() % 201u) - 100;
                slice.assign((size_t)SLICE_SAMPLES, c32(0.0f, 0.0f));
                for (int n = 0; n < FRAME_BODY; n++) {
                    const int at = SLICE_LEAD + toff + n;
                    if (at < 0 || at >= SLICE_SAMPLES) continue;
                    const double ph = 2.0 * 3.14159265358979323846 * cfo * (double)n;
                    const std::complex<double> v = std::complex<double>(src[n].real(), src[n].imag()) * std::complex<double>(std::cos(ph), std::sin(ph));
                    slice[(size_t)at] = c32((float)v.real(), (float)v.imag());
                }
                src = slice.data();
            }
            HIPCK(hipMemcpy(d_base + (size_t)d * in_samples, src, in_samples * sizeof(c32), hipMemcpyHostToDevice));
        }
        for (size_t e = 0; e < E; e++)                                  // ensemble e of worker `rank` carries multiplex (e + rank) mod distinct
            HIPCK(hipMemcpyAsync(d_iq + e * in_samples, d_base + ((e + (size_t)rank) % (size_t)n_distinct) * in_samples, in_samples * sizeof(c32), hipMemcpyDeviceToDevice, st[0]));
        HIPCK(hipStreamSynchronize(st[0]));
        HIPCK(hipFree(d_base));
    }
    dabgpu_sync_cfg sync_cfg;
    dabgpu_sync_cfg_default(&sync_cfg);
    std::vector<float*> d_corr((size_t)inflight); std::vector<uint8_t*> d_fib((size_t)inflight), d_msc((size_t)inflight);
    std::vector<dabgpu_codeword_result*> d_fres((size_t)inflight), d_mres((size_t)inflight);
    for (int k = 0; k < inflight; k++) {
        HIPCK(hipMalloc((void**)&d_corr[(size_t)k], E * 76 * 2 * sizeof(float)));
        HIPCK(hipMalloc((void**)&d_fib[(size_t)k], E * 4 * 96));
        HIPCK(hipMalloc((void**)&d_msc[(size_t)k], E * 4 * cif_out));
        HIPCK(hipMalloc((void**)&d_fres[(size_t)k], E * 4 * sizeof(dabgpu_codeword_result)));
        HIPCK(hipMalloc((void**)&d_mres[(size_t)k], E * 4 * N_SUB * sizeof(dabgpu_codeword_result)));
    }
    std::vector<dabgpu_subchannel> subs((size_t)N_SUB);
    for (int s = 0; s < N_SUB; s++) { subs[(size_t)s] = dabgpu_subchannel{}; subs[(size_t)s].start_address = SUB_CU * s; subs[(size_t)s].length = SUB_CU; subs[(size_t)s].eep_prot_level = 2; }

    // frame j runs on lane j mod inflight; msc(j) reads the ring slots of frames j-4..j (waits for demod(j-1), ...), demod(j) overwrites
    // the slot of frame j-H, last read by msc(j-H+4)
    const int NEV = 64;
    std::vector<hipEvent_t> ev_demod((size_t)NEV), ev_msc((size_t)NEV);
    for (int i = 0; i < NEV; i++) { HIPCK(hipEventCreateWithFlags(&ev_demod[(size_t)i], hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&ev_msc[(size_t)i], hipEventDisableTiming)); }
    long j = 0;
    auto step = [&]() -> bool {
        const int k = (int)(j % inflight), slot = (int)(j % H);
        hipStream_t s = st[(size_t)k];
        if (inflight > 1 && j - H + 4 >= 0) HIPCK(hipStreamWaitEvent(s, ev_msc[(size_t)((j - H + 4) % NEV)], 0));
        if (synced) {
            // frame j's synchroniser starts from the records frame j - 1's fine-frequency update left (one receiver's frames are a serial chain)
            if (inflight > 1 && j >= 1) HIPCK(hipStreamWaitEvent(s, ev_demod[(size_t)((j - 1) % NEV)], 0));
            DABCK(dabgpu_ofdm_sync_demod_frames(ctx[(size_t)k], reinterpret_cast<const float*>(d_iq), E, (size_t)SLICE_SAMPLES, (size_t)SLICE_LEAD, &sync_cfg, d_states,
                                                d_hist + (size_t)slot * FRAME_BITS, d_corr[(size_t)k], 0, stride, DABGPU_BITS_MSC_CLASSED, nullptr, s));
        } else {
            DABCK(dabgpu_ofdm_demod_frames_history(ctx[(size_t)k], d_iq, DABGPU_IQ_RAW_F32L, E, nullptr, d_hist + (size_t)slot * FRAME_BITS, d_corr[(size_t)k], 0, stride,
                                                   DABGPU_BITS_MSC_CLASSED, s));
        }
        HIPCK(hipEventRecord(ev_demod[(size_t)(j % NEV)], s));
        for (int d = 1; d < inflight; d++) if (j - d >= 0) HIPCK(hipStreamWaitEvent(s, ev_demod[(size_t)((j - d) % NEV)], 0));
        // FIC + MSC of the frame in one call: the FIB groups are decoded inside the MSC launch
        DABCK(dabgpu_decode_frames_layout(ctx[(size_t)k], d_hist, E, stride, H, slot, subs.data(), N_SUB, d_fib[(size_t)k], d_fres[(size_t)k], d_msc[(size_t)k],
                                          4 * cif_out, d_mres[(size_t)k], 0, DABGPU_BITS_MSC_CLASSED, s));
        HIPCK(hipEventRecord(ev_msc[(size_t)(j % NEV)], s));
        j++;
        return true;
    };
    // explicit calibration of the demodulator's run length for this batch size, once per context, before anything is timed
    // (symbols_per_block = 0 in step() resolves to what is recorded here; the data path never measures)
    for (int k = 0; k < inflight; k++) DABCK(dabgpu_ofdm_tune(ctx[(size_t)k], d_iq, DABGPU_IQ_RAW_F32L, E, d_hist, stride, DABGPU_BITS_MSC_CLASSED, synced ? 1 : 0, st[(size_t)k], &R->spb));
    for (int i = 0; i < H + inflight; i++) if (!step()) return false;   // fill the history ring: the time de-interleaver needs 16 CIFs
    HIPCK(hipDeviceSynchronize());
    bar->wait();                                                       // all workers start their timed steps together
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) {
        if (!step()) return false;
        if (i + 1 == probe_step) {                                     // every scratch buffer of the library has its final size long before this
            size_t total = 0;
            for (int k = 0; k < inflight; k++) HIPCK(hipStreamSynchronize(st[(size_t)k]));
            bar->wait();                                               // (all workers are past their allocations when any of them measures)
            HIPCK(hipMemGetInfo(&R->free_at_probe, &total));
            bar->wait();
        }
    }
    for (int k = 0; k < inflight; k++) HIPCK(hipStreamSynchronize(st[(size_t)k]));
    R->ms_per_step = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    if (probe_step > 0) { size_t total = 0; bar->wait(); HIPCK(hipMemGetInfo(&R->free_at_end, &total)); bar->wait(); }

    // ---- every byte of the last frame of every lane against what was transmitted ----
    std::vector<uint8_t> h_fib(E * 4 * 96), h_msc(E * 4 * cif_out);
    std::vector<dabgpu_codeword_result> h_fres(E * 4);
    uint64_t dig = 0xCBF29CE484222325ull;
    for (int k = 0; k < inflight; k++) {
        HIPCK(hipMemcpy(h_fib.data(), d_fib[(size_t)k], h_fib.size(), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(h_msc.data(), d_msc[(size_t)k], h_msc.size(), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(h_fres.data(), d_fres[(size_t)k], h_fres.size() * sizeof(dabgpu_codeword_result), hipMemcpyDeviceToHost));
        dig = fnv1a(fnv1a(dig, h_fib.data(), h_fib.size()), h_msc.data(), h_msc.size());
        for (size_t e = 0; e < E; e++) {
            const Multiplex& M = mux[(e + (size_t)rank) % (size_t)n_distinct];
            R->fib_mismatch += std::memcmp(&h_fib[e * 384], M.fibs.data(), 384) != 0;
            for (int c = 0; c < 4; c++) {
                R->msc_mismatch += std::memcmp(&h_msc[(e * 4 + (size_t)c) * cif_out], M.payload.data(), cif_out) != 0;
                R->crc_pass += __builtin_popcount(h_fres[e * 4 + (size_t)c].crc_ok_mask & 7u);
            }
        }
        R->crc_expected += (long)E * 12;
    }
    R->digest = dig;
    for (int k = 0; k < inflight; k++) {
        (void)hipFree(d_corr[(size_t)k]); (void)hipFree(d_fib[(size_t)k]); (void)hipFree(d_msc[(size_t)k]); (void)hipFree(d_fres[(size_t)k]); (void)hipFree(d_mres[(size_t)k]);
        dabgpu_destroy(ctx[(size_t)k]); (void)hipStreamDestroy(st[(size_t)k]);
    }
    for (int i = 0; i < NEV; i++) { (void)hipEventDestroy(ev_demod[(size_t)i]); (void)hipEventDestroy(ev_msc[(size_t)i]); }
    (void)hipFree(d_iq); (void)hipFree(d_hist); (void)hipFree(d_states);
    R->ok = true;
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<int> devices;
    size_t E = 8192;
    int steps = 10, distinct = 8, inflight = 2, probe_step = 0;
    bool identical = false, synced = true;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", k.c_str()); std::exit(2); } return argv[++a]; };
        if (k == "--devices") { const char* v = val(); for (const char* p = v; *p;) { devices.push_back((int)std::strtol(p, (char**)&p, 10)); if (*p == ',') p++; } }
        else if (k == "--ensembles") E = (size_t)std::atol(val());
        else if (k == "--steps") steps = std::atoi(val());
        else if (k == "--distinct") distinct = std::atoi(val());
        else if (k == "--inflight") inflight = std::atoi(val());
        else if (k == "--identical") identical = true;
        else if (k == "--aligned") synced = false;
        else if (k == "--mem-probe-step") probe_step = std::atoi(val());      // soak: device memory free after this step and after the last one
        else { std::fprintf(stderr, "usage: %s --devices 0,1,... [--ensembles E] [--steps K] [--distinct D] [--inflight 1|2] [--identical] [--aligned]\n", argv[0]); return 2; }
    }
    if (devices.empty()) devices.push_back(0);
    if (E == 0 || steps < 1 || distinct < 1 || inflight < 1 || inflight > 4 || probe_step < 0 || probe_step >= steps) { std::fprintf(stderr, "bad arguments\n"); return 2; }
    if (dabgpu_device_count() <= 0) { std::fprintf(stderr, "no gfx950 device (this path has no CPU fallback)\n"); return 1; }
    Barrier bar; bar.n = (int)devices.size();
    std::vector<Result> res(devices.size());
    std::vector<std::thread> th;
    std::atomic<int> failed{0};
    for (size_t r = 0; r < devices.size(); r++)
        th.emplace_back([&, r] {
            if (!worker(identical ? -1 : (int)r, devices[r], E, steps, distinct, inflight, probe_step, synced, &bar, &res[r])) {
                failed++;
                // a worker that failed before the start line must not leave the others waiting at it
                std::unique_lock<std::mutex> lk(bar.mu);
                bar.n--; if (bar.waiting == bar.n && bar.n > 0) { bar.waiting = 0; bar.gen++; bar.cv.notify_all(); }
            }
        });
    for (auto& t : th) t.join();
    double worst = 0.0;
    bool all_ok = failed == 0;
    for (const auto& r : res) { worst = std::max(worst, r.ms_per_step); all_ok = all_ok && r.ok && r.fib_mismatch == 0 && r.msc_mismatch == 0 && r.crc_pass == r.crc_expected; }
    std::printf("{\"program\": \"multi_gpu_harness\", \"workload\": \"BASELINE configs[4] per device: %s + FIC Viterbi + 18 x 48 CU EEP 3-A MSC per transmission frame\", "
                "\"workers\": %zu, \"ensembles_per_worker\": %zu, \"steps\": %d, \"frames_in_flight\": %d, \"distinct_multiplexes\": %d, "
                "\"scaling\": \"weak: independent ensembles per device, one host thread + contexts + streams per device, no collective\", "
                "\"frames_per_s\": %.1f, \"ms_per_step_slowest_worker\": %.4f, \"all_outputs_equal_transmitted\": %s, \"mem_probe_step\": %d, \"per_worker\": [",
                synced ? "PRS synchronisation + OFDM demod at the tracked offsets + fine-frequency update (per-multiplex carrier and timing offsets)" : "OFDM demod of frame-aligned input (synchronisation bypassed)",
                devices.size(), E, steps, inflight, distinct, all_ok && worst > 0.0 ? (double)devices.size() * (double)E / worst * 1e3 : 0.0, worst, all_ok ? "true" : "false", probe_step);
    for (size_t r = 0; r < res.size(); r++)
        std::printf("%s{\"device\": %d, \"ok\": %s, \"ms_per_step\": %.4f, \"fib_groups_wrong\": %ld, \"msc_cifs_wrong\": %ld, \"fib_crc_pass\": %ld, \"fib_crc_expected\": %ld, \"digest\": \"%016llx\", "
                    "\"symbols_per_block\": %d, \"device_free_bytes_at_probe\": %zu, \"device_free_bytes_at_end\": %zu}",
                    r ? ", " : "", res[r].device, res[r].ok ? "true" : "false", res[r].ms_per_step, res[r].fib_mismatch, res[r].msc_mismatch, res[r].crc_pass, res[r].crc_expected,
                    (unsigned long long)res[r].digest, res[r].spb, res[r].free_at_probe, res[r].free_at_end);
    std::printf("]}\n");
    return all_ok ? 0 : 1;
}
