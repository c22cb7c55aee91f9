// mirror_harness.cpp -- drives the C++ mirror classes the way basic_radio_app does (examples/basic_radio_app.cpp:404-419,
// src/basic_radio/basic_radio.cpp:41-65, basic_fic_runner.cpp:34-49, basic_dab_plus_channel.cpp:47-51): raw IQ file ->
// OFDM_Demod::Process in blocks -> frame bits -> FIC_Decoder x4 + MSC_Decoder x4 per sub-channel.  Everything it
// produces is written to files that tests/test_gpu_cpp_mirror.py compares byte for byte with the CPU oracle.
//
//   mirror_harness <iq.c32> <out_dir> <block_size> [<start_cu> <length_cu> <eep_level> <eep_type_b>]...
// DABGPU_HARNESS_THREADS=n: the sub-channels of a CIF are decoded by n threads (s = t, t + n, ...), the way basic_radio's thread pool
// runs one task per sub-channel (src/basic_radio/basic_radio.cpp:51-62); outputs are identical, written in sub-channel order.
// DABGPU_HARNESS_BENCH=1: nothing is written; the harness times the whole run (frames/s against the 10.42 frames/s of a live
// signal) and every DecodeFIBGroup / DecodeCIF call (one synchronous launch + two copies each) and prints one JSON line.
#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "dab/algorithms/dab_viterbi_decoder.h"
#include "dab/constants/dab_parameters.h"
#include "dab/constants/puncture_codes.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/cif_deinterleaver.h"
#include "dab/msc/msc_decoder.h"
#include "ofdm/ofdm_helpers.h"

static bool g_bench = false;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void append(const std::string& path, const void* p, size_t n) {
    if (g_bench) return;
    std::ofstream f(path, std::ios::binary | std::ios::app);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)n);
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s iq.c32 out_dir block_size [start len level type_b]...\n", argv[0]); return 2; }
    const std::string out = argv[2];
    const size_t block = (size_t)std::atol(argv[3]);
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }

    g_bench = std::getenv("DABGPU_HARNESS_BENCH") != nullptr;
    const char* thr_env = std::getenv("DABGPU_HARNESS_THREADS");
    const int n_threads = thr_env ? std::atoi(thr_env) : 1;
    std::vector<double> lat_fib, lat_cif;
    const DAB_Parameters dab = get_dab_parameters(1);
    const char* mode_env = std::getenv("DABGPU_HARNESS_MODE");
    const int tx_mode = mode_env ? std::atoi(mode_env) : 1;
    auto demod = Create_OFDM_Demodulator(tx_mode);
    if (const char* thr = std::getenv("DABGPU_HARNESS_PEAK_DB")) demod->GetConfig().sync.impulse_peak_threshold_db = (float)std::atof(thr);
    demod->EnableDebugBuffers(true);
    FIC_Decoder fic((size_t)dab.nb_fib_cif_bits, (size_t)dab.nb_fibs_per_cif);
    fic.OnFIB().Attach([&](tcb::span<const uint8_t> fib) { append(out + "/fibs.bin", fib.data(), fib.size()); });

    std::vector<std::unique_ptr<MSC_Decoder>> msc;
    std::vector<Subchannel> subs;
    for (int a = 4; a + 3 < argc; a += 4) {
        Subchannel sc((subchannel_id_t)subs.size());
        sc.start_address = (subchannel_addr_t)std::atoi(argv[a]);
        sc.length = (subchannel_size_t)std::atoi(argv[a + 1]);
        sc.eep_prot_level = (eep_protection_level_t)std::atoi(argv[a + 2]);
        sc.eep_type = std::atoi(argv[a + 3]) ? EEP_Type::TYPE_B : EEP_Type::TYPE_A;
        sc.is_complete = true;
        subs.push_back(sc);
        msc.push_back(std::make_unique<MSC_Decoder>(sc));
    }
    std::unique_ptr<CIF_Deinterleaver> deint;
    if (!subs.empty()) deint = std::make_unique<CIF_Deinterleaver>(subs[0].length * 8);
    DAB_Viterbi_Decoder vit;
    vit.set_traceback_length(768);

    int n_frames = 0;
    demod->On_OFDM_Frame().Attach([&](tcb::span<const viterbi_bit_t> bits) {
        n_frames++;
        append(out + "/frame_bits.bin", bits.data(), bits.size());
        const float st[4] = {demod->GetCoarseFrequencyOffset(), demod->GetFineFrequencyOffset(), (float)demod->GetFineTimeOffset(),
                             (float)demod->GetTotalFramesDesync()};
        append(out + "/states.bin", st, sizeof(st));
        const size_t nb_fft = demod->GetOFDMParams().nb_fft;
        auto fft = demod->GetFrameFFT();
        append(out + "/fft_sym1.bin", fft.data() + nb_fft, nb_fft * sizeof(std::complex<float>));
        if (tx_mode != 1) return;                       // the DAB layer above the soft bits is mode I only (fic_decoder.cpp:61-72)
        auto dq = demod->GetFrameDataVec();
        append(out + "/dqpsk_sym0.bin", dq.data(), 1536 * sizeof(std::complex<float>));
        auto fic_bits = bits.subspan(0, (size_t)dab.nb_fic_bits);
        auto msc_bits = bits.subspan((size_t)dab.nb_fic_bits, (size_t)dab.nb_msc_bits);
        for (int c = 0; c < dab.nb_cifs; c++) {                                         // basic_fic_runner.cpp:44-48
            auto grp = fic_bits.subspan((size_t)c * dab.nb_fib_cif_bits, (size_t)dab.nb_fib_cif_bits);
            const double t_fib = now_us();
            fic.DecodeFIBGroup(grp, (size_t)c);
            lat_fib.push_back(now_us() - t_fib);
            const uint32_t m = fic.GetLastCrcMask();
            const uint64_t e = fic.GetLastPathError();
            append(out + "/fic_status.bin", &m, 4);
            append(out + "/fic_status.bin", &e, 8);
            if (c == 0 && !g_bench) {                                                   // DAB_Viterbi_Decoder used directly, fic_decoder.cpp:74-87
                uint8_t raw[96];
                vit.reset();
                size_t used = vit.update(grp, GetPunctureCode(16), 128 * 21);
                used += vit.update(grp.subspan(used), GetPunctureCode(15), 128 * 3);
                used += vit.update(grp.subspan(used), PI_X, 24);
                const uint64_t err = vit.chainback(raw);
                append(out + "/viterbi_raw.bin", raw, 96);
                const uint64_t meta[3] = {used, vit.get_current_decoded_bit(), err};
                append(out + "/viterbi_meta.bin", meta, sizeof(meta));
            }
        }
        for (int c = 0; c < dab.nb_cifs; c++) {                                         // basic_dab_plus_channel.cpp:47-51
            auto cif = msc_bits.subspan((size_t)c * dab.nb_cif_bits, (size_t)dab.nb_cif_bits);
            std::vector<tcb::span<uint8_t>> got(msc.size());
            std::vector<double> took(msc.size(), 0.0);
            auto work = [&](size_t first, size_t step) {
                for (size_t s = first; s < msc.size(); s += step) {
                    const double t_cif = now_us();
                    got[s] = msc[s]->DecodeCIF(cif);
                    took[s] = now_us() - t_cif;
                }
            };
            if (n_threads <= 1) {
                work(0, 1);
            } else {
                std::vector<std::thread> pool;
                for (int t = 0; t < n_threads; t++) pool.emplace_back(work, (size_t)t, (size_t)n_threads);
                for (auto& th : pool) th.join();
            }
            for (size_t s = 0; s < msc.size(); s++) {
                lat_cif.push_back(took[s]);
                const uint32_t nb = (uint32_t)got[s].size();
                append(out + "/msc_" + std::to_string(s) + ".bin", &nb, 4);
                append(out + "/msc_" + std::to_string(s) + ".bin", got[s].data(), got[s].size());
            }
            if (deint && !g_bench) {
                const size_t nbits = (size_t)subs[0].length * 64;
                std::vector<viterbi_bit_t> lf(nbits);
                deint->Consume(cif.subspan((size_t)subs[0].start_address * 64, nbits));
                const uint8_t ok = deint->Deinterleave(lf) ? 1 : 0;
                append(out + "/deint.bin", &ok, 1);
                if (ok) append(out + "/deint.bin", lf.data(), nbits);
            }
        }
    });

    if (g_bench) demod->EnableDebugBuffers(false);
    // DABGPU_HARNESS_SCHEDULE = a text file of block lengths, one per Process() call (the last one repeats; a NEGATIVE entry -n = the caller
    // calls Reset() and then hands n samples): the block size of argv is used without it
    std::vector<long> schedule;
    if (const char* sp = std::getenv("DABGPU_HARNESS_SCHEDULE")) {
        std::ifstream sf(sp);
        long v;
        while (sf >> v) if (v != 0) schedule.push_back(v);
    }
    size_t max_block = block;
    for (long v : schedule) max_block = std::max(max_block, (size_t)std::labs(v));
    std::vector<std::complex<float>> buf(max_block);
    // DABGPU_HARNESS_LOOPS = n (timing mode): the capture n times back to back -- the start of a stream (first launches, first-use allocations,
    // first touch of every page-locked buffer, the decoders' 16-CIF run-in) is then a small part of the run
    const int loops = g_bench && std::getenv("DABGPU_HARNESS_LOOPS") ? std::max(1, std::atoi(std::getenv("DABGPU_HARNESS_LOOPS"))) : 1;
    const double t_run = now_us();
    size_t call = 0;
    for (int loop = 0; loop < loops; loop++) {
        if (loop) { in.clear(); in.seekg(0); }
        for (; in; call++) {
            long want = schedule.empty() ? (long)block : schedule[std::min(call, schedule.size() - 1)];
            if (want < 0) { demod->Reset(); want = -want; }
            in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)((size_t)want * sizeof(std::complex<float>)));
            const size_t got = (size_t)in.gcount() / sizeof(std::complex<float>);
            if (got == 0) break;
            demod->Process(tcb::span<const std::complex<float>>(buf.data(), got));
        }
    }
    demod->Synchronize();                           // every frame handed to the device has come out of the observers
    if (g_bench) {
        const double sec = (now_us() - t_run) * 1e-6;
        auto stats = [](std::vector<double>& v, double& med, double& p99, double& mx) {
            med = p99 = mx = 0;
            if (v.empty()) return;
            std::sort(v.begin(), v.end());
            med = v[v.size() / 2]; p99 = v[(size_t)((double)(v.size() - 1) * 0.99)]; mx = v.back();
        };
        double fm, fp, fx, cm, cp, cx;
        stats(lat_fib, fm, fp, fx); stats(lat_cif, cm, cp, cx);
        std::printf("{\"frames\": %d, \"seconds\": %.4f, \"frames_per_s\": %.2f, \"x_realtime\": %.2f, \"sub_channels\": %zu, \"decode_threads\": %d, "
                    "\"decode_fib_group_us\": {\"median\": %.1f, \"p99\": %.1f, \"max\": %.1f, \"calls\": %zu}, "
                    "\"decode_cif_us\": {\"median\": %.1f, \"p99\": %.1f, \"max\": %.1f, \"calls\": %zu}}\n",
                    n_frames, sec, n_frames / sec, n_frames / sec / (2.048e6 / 196608.0), msc.size(), n_threads, fm, fp, fx, lat_fib.size(), cm, cp, cx, lat_cif.size());
        return 0;
    }
    std::printf("frames=%d read=%d desync=%d state=%d signal_avg=%.9g\n", n_frames, demod->GetTotalFramesRead(),
                demod->GetTotalFramesDesync(), (int)demod->GetState(), demod->GetSignalAverage());
    return 0;
}
