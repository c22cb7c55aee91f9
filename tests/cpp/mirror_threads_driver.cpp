// mirror_threads_driver.cpp -- the C++ mirror classes (dab-radio_amd/host/**) driven the way basic_radio drives the reference's, with
// SEVERAL receivers in one process and the decoders on their own threads, for ThreadSanitizer / AddressSanitizer runs on a machine
// without a GPU: linked against tests/cpp/fake_dabgpu_oracle.cpp (the C ABI on the CPU oracle, test infrastructure) instead of
// libdabgpu.so.  What is under test is the host code's threading: the process-wide frame batcher with one session per demodulator, the
// shared context, the decoders' registration / matching / fall-back logic.
//
//   mirror_threads_driver <block> <start_cu> <len_cu> <eep_level> <eep_type_b> [more sub-channels ...] -- <iq0.c32> <iq1.c32> ...
// Phase 1: every receiver alone, everything on one thread.  Phase 2: all receivers at once -- per receiver one reader thread
// (OFDM_Demod::Process) and one radio thread that takes the frames from a queue (the reference's ring buffer between its OFDM and radio
// threads, examples/app_helpers/app_ofdm_blocks.h:32-35) and decodes the FIC on itself and the sub-channels on two workers.  The bytes
// of phase 2 must equal those of phase 1, receiver by receiver; prints one JSON line; exit status 0 only then.
#include <atomic>
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <dirent.h>
#include <dlfcn.h>
#include <malloc.h>
#include <fstream>
#include <map>
#include <pthread.h>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dab/constants/dab_parameters.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/msc_decoder.h"
#include "ofdm/ofdm_helpers.h"

namespace {

struct Output { std::vector<uint8_t> fibs, msc; int frames = 0; int cifs_out = 0; bool keep = true; size_t n_fib_bytes = 0, n_msc_bytes = 0; };   // keep = false (timing mode): count, do not collect

uint64_t fnv(const std::vector<uint8_t>& v, uint64_t h = 0xCBF29CE484222325ull) { for (uint8_t b : v) { h ^= b; h *= 0x100000001B3ull; } return h; }

std::vector<std::complex<float>> load(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    const size_t n = (size_t)f.tellg() / sizeof(std::complex<float>);
    std::vector<std::complex<float>> v(n);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(std::complex<float>)));
    return v;
}

// the radio side of one receiver: FIC decoder + one MSC decoder per sub-channel
struct Radio {
    DAB_Parameters dab = get_dab_parameters(1);
    FIC_Decoder fic;
    std::vector<std::unique_ptr<MSC_Decoder>> msc;
    Output* out;
    Radio(const std::vector<Subchannel>& subs, Output* o) : fic((size_t)dab.nb_fib_cif_bits, (size_t)dab.nb_fibs_per_cif), out(o) {
        for (const auto& s : subs) msc.push_back(std::make_unique<MSC_Decoder>(s));
        fic.OnFIB().Attach([this](tcb::span<const uint8_t> fib) { out->n_fib_bytes += fib.size(); if (out->keep) out->fibs.insert(out->fibs.end(), fib.begin(), fib.end()); });
    }
    void frame(const std::vector<viterbi_bit_t>& bits, bool workers) {
        out->frames++;
        tcb::span<const viterbi_bit_t> all(bits.data(), bits.size());
        auto fic_bits = all.subspan(0, (size_t)dab.nb_fic_bits);
        auto msc_bits = all.subspan((size_t)dab.nb_fic_bits, (size_t)dab.nb_msc_bits);
        for (int c = 0; c < dab.nb_cifs; c++) fic.DecodeFIBGroup(fic_bits.subspan((size_t)c * dab.nb_fib_cif_bits, (size_t)dab.nb_fib_cif_bits), (size_t)c);
        for (int c = 0; c < dab.nb_cifs; c++) {
            auto cif = msc_bits.subspan((size_t)c * dab.nb_cif_bits, (size_t)dab.nb_cif_bits);
            std::vector<std::vector<uint8_t>> got(msc.size());
            auto work = [&](size_t first, size_t step) {
                for (size_t s = first; s < msc.size(); s += step) { auto r = msc[s]->DecodeCIF(cif); got[s].assign(r.begin(), r.end()); }
            };
            if (workers && msc.size() > 1) { std::thread a(work, 0, 2), b(work, 1, 2); a.join(); b.join(); }
            else work(0, 1);
            for (auto& g : got) { if (!g.empty()) out->cifs_out++; out->n_msc_bytes += g.size(); if (out->keep) out->msc.insert(out->msc.end(), g.begin(), g.end()); }
        }
    }
};

struct FrameQueue {
    std::mutex mu; std::condition_variable cv, cv_space; std::deque<std::vector<viterbi_bit_t>> q; bool done = false;
    // two frames deep and blocking when full, like the reference app's ring buffer between its OFDM and radio threads
    // (ThreadedRingBuffer<viterbi_bit_t>(nb_frame_bits * 2), examples/basic_radio_app.cpp:320): the radio side never falls more than two
    // frames behind, well inside the 8 frames whose decoded bytes a demodulator's frame session keeps
    void push(tcb::span<const viterbi_bit_t> b) {
        { std::unique_lock<std::mutex> g(mu); cv_space.wait(g, [&] { return q.size() < 2; }); q.emplace_back(b.begin(), b.end()); }
        cv.notify_one();
    }
    bool pop(std::vector<viterbi_bit_t>& out) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done || !q.empty(); });
        if (q.empty()) return false;
        out = std::move(q.front()); q.pop_front();
        lk.unlock();
        cv_space.notify_one();
        return true;
    }
    void finish() { { std::lock_guard<std::mutex> g(mu); done = true; } cv.notify_all(); }
};

// development (DABGPU_DRIVER_CPU=1, timing mode): CPU time of every thread of the process by thread name, read from /proc while all of them are alive
void print_cpu_by_thread_name(size_t frames) {
    std::map<std::string, std::pair<double, int>> by_name;
    if (DIR* d = opendir("/proc/self/task")) {
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            const std::string base = std::string("/proc/self/task/") + e->d_name;
            std::ifstream fc(base + "/comm"), fs(base + "/schedstat");
            std::string name; std::getline(fc, name);
            double ns = 0.0; fs >> ns;
            by_name[name].first += ns; by_name[name].second++;
        }
        closedir(d);
    }
    std::fprintf(stderr, "CPU by thread name, us per frame over %zu frames:", frames);
    double total = 0.0;
    for (const auto& kv : by_name) { std::fprintf(stderr, " %s[%d] %.1f,", kv.first.c_str(), kv.second.second, kv.second.first / 1e3 / (double)frames); total += kv.second.first; }
    std::fprintf(stderr, " total %.1f\n", total / 1e3 / (double)frames);
}

// development (DABGPU_DRIVER_LOOPS=n, timing mode): resident set of the process and used device memory (when the HIP runtime is in the process)
std::string memory_sample(int loop) {
    double rss_mb = 0.0;
    { std::ifstream f("/proc/self/status"); std::string ln; while (std::getline(f, ln)) if (ln.rfind("VmRSS:", 0) == 0) rss_mb = std::atof(ln.c_str() + 6) / 1024.0; }
    double dev_mb = -1.0;
    using mem_fn = int (*)(size_t*, size_t*);
    if (auto fn = reinterpret_cast<mem_fn>(dlsym(RTLD_DEFAULT, "hipMemGetInfo"))) { size_t fr = 0, tot = 0; if (fn(&fr, &tot) == 0) dev_mb = (double)(tot - fr) / 1048576.0; }
    // where the resident set lives: the C heap's live bytes (mallinfo2: arenas + mmapped blocks), anonymous pages as a whole, shared / device-mapped pages
    double anon_mb = 0.0, shmem_mb = 0.0, file_mb = 0.0;
    { std::ifstream f("/proc/self/status"); std::string ln;
      while (std::getline(f, ln)) {
          if (ln.rfind("RssAnon:", 0) == 0) anon_mb = std::atof(ln.c_str() + 8) / 1024.0;
          if (ln.rfind("RssShmem:", 0) == 0) shmem_mb = std::atof(ln.c_str() + 9) / 1024.0;
          if (ln.rfind("RssFile:", 0) == 0) file_mb = std::atof(ln.c_str() + 8) / 1024.0;
      } }
    using dump_fn = void (*)(const char*);                   // tools/exp/leakhist.c preloaded: live heap blocks by size
    if (auto fn = reinterpret_cast<dump_fn>(dlsym(RTLD_DEFAULT, "leakhist_dump"))) fn(("loop " + std::to_string(loop)).c_str());
    const struct mallinfo2 mi = mallinfo2();
    char buf[400];
    std::snprintf(buf, sizeof(buf), "{\"loop\": %d, \"host_rss_MB\": %.1f, \"rss_anon_MB\": %.1f, \"rss_shmem_MB\": %.1f, \"rss_file_MB\": %.1f, \"heap_in_use_MB\": %.1f, "
                  "\"heap_arenas_MB\": %.1f, \"device_used_MB\": %.1f}", loop, rss_mb, anon_mb, shmem_mb, file_mb, (double)(mi.uordblks + mi.hblkhd) / 1048576.0,
                  (double)(mi.arena + mi.hblkhd) / 1048576.0, dev_mb);
    return buf;
}

void feed(OFDM_Demod& demod, const std::vector<std::complex<float>>& iq, size_t block) {
    for (size_t k = 0; k < iq.size(); k += block)
        demod.Process(tcb::span<const std::complex<float>>(iq.data() + k, std::min(block, iq.size() - k)));
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s block start len level type_b [...] -- iq0.c32 [iq1.c32 ...]\n", argv[0]); return 2; }
    const size_t block = (size_t)std::atol(argv[1]);
    std::vector<Subchannel> subs;
    int a = 2;
    for (; a + 3 < argc && std::strcmp(argv[a], "--") != 0; a += 4) {
        Subchannel sc((subchannel_id_t)subs.size());
        sc.start_address = (subchannel_addr_t)std::atoi(argv[a]);
        sc.length = (subchannel_size_t)std::atoi(argv[a + 1]);
        sc.eep_prot_level = (eep_protection_level_t)std::atoi(argv[a + 2]);
        sc.eep_type = std::atoi(argv[a + 3]) ? EEP_Type::TYPE_B : EEP_Type::TYPE_A;
        sc.is_complete = true;
        subs.push_back(sc);
    }
    if (a >= argc || std::strcmp(argv[a], "--") != 0) { std::fprintf(stderr, "missing --\n"); return 2; }
    std::vector<std::vector<std::complex<float>>> iq;
    for (a++; a < argc; a++) iq.push_back(load(argv[a]));
    const size_t R = iq.size();

    // DABGPU_DRIVER_BENCH=1 (tools/bench_mirror_multi.py): only phase 2, decoders on the radio thread itself, timed -- R receivers in one
    // process, each with its reader thread, its radio thread and its demodulator's delivery thread
    const bool bench = std::getenv("DABGPU_DRIVER_BENCH") != nullptr;
    // ---- phase 1: one receiver at a time, one thread ----
    std::vector<Output> serial(R);
    for (size_t r = 0; r < R && !bench; r++) {
        auto demod = Create_OFDM_Demodulator(1);
        Radio radio(subs, &serial[r]);
        demod->On_OFDM_Frame().Attach([&](tcb::span<const viterbi_bit_t> bits) { radio.frame(std::vector<viterbi_bit_t>(bits.begin(), bits.end()), false); });
        feed(*demod, iq[r], block);
        demod->Synchronize();                                               // the last frames come out of the delivery thread
    }

    // ---- phase 2: all receivers at once; reader thread + radio thread (+ two decode workers) per receiver ----
    std::vector<Output> threaded(R);
    // start line: every receiver's objects exist (18 decoders = 18 device contexts each) before the first block is fed / the clock starts
    std::mutex start_mu; std::condition_variable start_cv; size_t ready = 0, finished = 0;
    const bool cpu_report = bench && std::getenv("DABGPU_DRIVER_CPU") != nullptr;
    const int loops = bench && std::getenv("DABGPU_DRIVER_LOOPS") ? std::max(1, std::atoi(std::getenv("DABGPU_DRIVER_LOOPS"))) : 1;
    std::string memory_samples;
    auto t_start = std::chrono::steady_clock::now();
    auto start_line = [&] {
        std::unique_lock<std::mutex> lk(start_mu);
        if (++ready == 2 * R) { t_start = std::chrono::steady_clock::now(); start_cv.notify_all(); }
        else start_cv.wait(lk, [&] { return ready == 2 * R; });
    };
    {
        std::vector<std::unique_ptr<FrameQueue>> queues;
        std::vector<std::thread> threads;
        for (size_t r = 0; r < R; r++) queues.push_back(std::make_unique<FrameQueue>());
        for (size_t r = 0; r < R; r++) {
            threads.emplace_back([&, r] {                                   // radio thread: owns the decoders of receiver r
                pthread_setname_np(pthread_self(), "drv-radio");
                threaded[r].keep = !bench;
                Radio radio(subs, &threaded[r]);
                start_line();
                std::vector<viterbi_bit_t> bits;
                while (queues[r]->pop(bits)) radio.frame(bits, !bench);
            });
            threads.emplace_back([&, r] {                                   // reader thread: owns the demodulator of receiver r
                pthread_setname_np(pthread_self(), "drv-reader");
                auto demod = Create_OFDM_Demodulator(1);
                demod->On_OFDM_Frame().Attach([&](tcb::span<const viterbi_bit_t> bits) { queues[r]->push(bits); });
                start_line();
                // (DABGPU_DRIVER_LOOPS: the capture again and again -- every wrap breaks the framing: loss of lock, reset, re-acquisition with frames in flight)
                for (int loop = 0; loop < loops; loop++) {
                    feed(*demod, iq[r], block);
                    if (r == 0 && loops > 1 && (loop == 0 || loop == loops - 1 || (loops >= 8 && loop % (loops / 8) == 0) || (loops < 8 && (loop == loops / 4 || loop == loops / 2)))) {
                        std::lock_guard<std::mutex> lk(start_mu);
                        memory_samples += (memory_samples.empty() ? "" : ", ") + memory_sample(loop);
                    }
                }
                demod->Synchronize();
                if (cpu_report) {                                           // every thread of every receiver is still alive at this line
                    std::unique_lock<std::mutex> lk(start_mu);
                    if (++finished == R) {
                        size_t frames = 0;
                        for (size_t k = 0; k < R; k++) frames += (size_t)threaded[k].frames;
                        print_cpu_by_thread_name(frames);
                        start_cv.notify_all();
                    } else start_cv.wait(lk, [&] { return finished == R; });
                }
                queues[r]->finish();
            });
        }
        for (auto& t : threads) t.join();
    }

    if (bench) {
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        size_t frames = 0, bytes = 0;
        for (size_t r = 0; r < R; r++) { frames += (size_t)threaded[r].frames; bytes += threaded[r].n_msc_bytes + threaded[r].n_fib_bytes; }
        std::printf("{\"receivers\": %zu, \"sub_channels\": %zu, \"frames\": %zu, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"x_realtime_per_receiver\": %.1f, "
                    "\"decoded_bytes\": %zu, \"loops\": %d, \"memory\": [%s]}\n", R, subs.size(), frames, sec, frames / sec, frames / sec / (double)R / (2.048e6 / 196608.0), bytes,
                    loops, memory_samples.c_str());
        return frames > 0 ? 0 : 1;
    }
    bool ok = true;
    std::printf("{\"receivers\": %zu, \"sub_channels\": %zu, \"per_receiver\": [", R, subs.size());
    for (size_t r = 0; r < R; r++) {
        const bool same = serial[r].fibs == threaded[r].fibs && serial[r].msc == threaded[r].msc && serial[r].frames == threaded[r].frames;
        ok = ok && same && serial[r].frames > 0;
        if (!same) if (const char* d = std::getenv("DABGPU_DRIVER_DUMP")) {      // development: what differs
            auto dump = [&](const char* name, const std::vector<uint8_t>& v) { std::ofstream f(std::string(d) + "/" + name + std::to_string(r) + ".bin", std::ios::binary); f.write((const char*)v.data(), (std::streamsize)v.size()); };
            dump("serial_fibs_", serial[r].fibs); dump("serial_msc_", serial[r].msc); dump("threaded_fibs_", threaded[r].fibs); dump("threaded_msc_", threaded[r].msc);
        }
        std::printf("%s{\"frames\": %d, \"fib_bytes\": %zu, \"msc_bytes\": %zu, \"cifs_with_output\": %d, \"digest\": \"%016llx\", \"threaded_equals_serial\": %s}", r ? ", " : "",
                    serial[r].frames, serial[r].fibs.size(), serial[r].msc.size(), serial[r].cifs_out, (unsigned long long)fnv(serial[r].msc, fnv(serial[r].fibs)), same ? "true" : "false");
    }
    std::printf("], \"ok\": %s}\n", ok ? "true" : "false");
    return ok ? 0 : 1;
}
