// host_logic_fuzz.cpp -- the device-free part of libdabgpu.so (dab-radio_amd/csrc/dabgpu_host_logic.cpp) under
// -fsanitize=address,undefined with fuzzed arguments: sub-channel descriptors (start + length beyond 864 CU, length 0 / negative,
// UEP index outside 0..63, more than 64 sub-channels), wav images with lying chunk sizes and truncations, codeword descriptors, the
// mapping cost model at degenerate sizes, table generators at invalid modes.  Every call must come back with a status (never crash,
// never read or write outside its arguments -- the sanitizers abort the process otherwise), and what it accepts must be consistent.
//
//   host_logic_fuzz [iterations] [seed]          built and run by tests/test_host_sanitizers.py
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "dabgpu_host_logic.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); g_fail++; } } while (0)

static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
static void put16(std::vector<uint8_t>& v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
static void tag(std::vector<uint8_t>& v, const char* t) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)t[i]); }

// a well-formed wav image the mutations start from (app_wav_reader.h:107-255)
static std::vector<uint8_t> wav_image(std::mt19937& rng) {
    static const uint16_t codes[5] = {1, 3, 6, 7, 0xFFFE};
    const uint16_t code = codes[rng() % 5];
    const uint16_t bits = (uint16_t)((code == 1) ? (uint16_t[]){8, 16, 24, 32}[rng() % 4] : (code == 3) ? (uint16_t[]){32, 64}[rng() % 2] : (code == 0xFFFE ? 16 : 8));
    const uint32_t fmt_size = code == 0xFFFE ? 40 : (rng() % 3 == 0 ? 18 : 16);
    std::vector<uint8_t> v;
    tag(v, "RIFF"); put32(v, 0); tag(v, "WAVE");
    tag(v, "fmt "); put32(v, fmt_size);
    put16(v, code); put16(v, (uint16_t)(1 + rng() % 2)); put32(v, 2048000); put32(v, 2048000u * 2 * bits / 8); put16(v, (uint16_t)(2 * bits / 8)); put16(v, bits);
    if (fmt_size == 18) put16(v, 0);
    if (fmt_size == 40) {
        put16(v, 22); put16(v, bits); put32(v, 3); put16(v, 1);
        static const uint8_t GUID[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
        v.insert(v.end(), GUID, GUID + 14);
    }
    if (code != 1 && code != 0xFFFE) { tag(v, "fact"); put32(v, 4); put32(v, 1000); }
    for (unsigned k = rng() % 3; k > 0; k--) { tag(v, "LIST"); const uint32_t n = rng() % 40; put32(v, n); v.insert(v.end(), n, (uint8_t)0x55); }
    tag(v, "data"); put32(v, 4000);
    v.insert(v.end(), 64, (uint8_t)0x80);
    return v;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
    std::mt19937 rng(argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u);
    auto pick = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
    auto wild = [&]() -> int {                     // mostly small, sometimes extreme
        switch (rng() % 8) { case 0: return INT32_MIN; case 1: return INT32_MAX; case 2: return -1; case 3: return 0; case 4: return pick(-70000, 70000); default: return pick(-5, 900); }
    };
    long accepted_plans = 0, accepted_wav = 0, accepted_cw = 0;

    for (int it = 0; it < iters; it++) {
        // ---- sub-channel plans ----
        {
            const int n_sub = (it % 97 == 0) ? pick(60, 80) : pick(0, 20);
            std::vector<dabgpu_subchannel> subs((size_t)(n_sub > 0 ? n_sub : 0));
            for (auto& s : subs) {
                const bool sane = rng() % 3 != 0;
                s.is_uep = (int)(rng() % 2);
                s.uep_prot_index = sane ? pick(0, 63) : wild();
                s.eep_prot_level = sane ? pick(0, 3) : wild();
                s.eep_type = sane ? pick(0, 1) : wild();
                s.start_address = sane ? pick(0, 800) : wild();
                s.length = sane ? (s.eep_type == 0 ? (int[]){12, 8, 6, 4}[s.eep_prot_level & 3] : (int[]){27, 21, 18, 15}[s.eep_prot_level & 3]) * pick(1, 8) : wild();
            }
            std::vector<dabgpu_msc_plan> plans;
            uint32_t off = 0, ms = 0, mo = 0;
            const int st = dabgpu_host_build_msc_plans(subs.empty() ? nullptr : subs.data(), n_sub, plans, &off, &ms, &mo);
            CHECK(st == DABGPU_OK || st == DABGPU_ERR_INVALID_ARG, "build_msc_plans status %d", st);
            if (st == DABGPU_OK) {
                accepted_plans++;
                CHECK((int)plans.size() == n_sub && n_sub <= 64, "plans %zu for %d sub-channels", plans.size(), n_sub);
                uint32_t run = 0;
                for (int k = 0; k < n_sub; k++) {
                    const dabgpu_msc_plan& P = plans[(size_t)k];
                    uint32_t steps = 0;
                    for (int j = 0; j < 4; j++) { steps += P.seg_steps[j]; CHECK(P.seg_steps[j] % 32 == 0 && P.seg_pi[j] <= 24, "segment %u x PI %u", P.seg_steps[j], P.seg_pi[j]); }
                    CHECK(P.n_steps == steps + 6 && P.out_offset == run && P.n_out_bytes * 8 == steps, "plan %d inconsistent", k);
                    CHECK(subs[(size_t)k].start_address >= 0 && subs[(size_t)k].start_address + subs[(size_t)k].length <= 864 && subs[(size_t)k].length > 0, "accepted a sub-channel outside the CIF");
                    CHECK(dabgpu_vit_in_bytes(P.seg_pi, P.seg_steps) <= (uint32_t)subs[(size_t)k].length * 64u, "plan %d consumes %u soft bits of %d CU", k,
                          dabgpu_vit_in_bytes(P.seg_pi, P.seg_steps), subs[(size_t)k].length);
                    run += P.n_out_bytes;
                    CHECK(P.n_steps <= ms && P.n_out_bytes <= mo, "maxima");
                }
                CHECK(run == off, "output bytes per CIF");
            }
            for (const auto& s : subs) {           // the public single-profile form on the same descriptors
                int pi[4], lx[4], nb = -7;
                const int nseg = dabgpu_subchannel_plan(&s, pi, lx, &nb);
                CHECK(nseg == -1 || nseg == 2 || nseg == 4, "subchannel_plan returned %d", nseg);
                if (nseg > 0) for (int j = 0; j < 4; j++) CHECK(lx[j] >= 0 && pi[j] >= 0 && pi[j] <= 24, "plan values");
            }
            (void)dabgpu_subchannel_plan(nullptr, nullptr, nullptr, nullptr);
        }
        // ---- wav headers ----
        {
            std::vector<uint8_t> img = (it % 5 == 0) ? std::vector<uint8_t>((size_t)pick(0, 200)) : wav_image(rng);
            if (it % 5 == 0) for (auto& b : img) b = (uint8_t)rng();
            for (unsigned m = rng() % 4; m > 0 && !img.empty(); m--) {
                const size_t at = rng() % img.size();
                switch (rng() % 4) {
                case 0: img[at] = (uint8_t)rng(); break;
                case 1: if (at + 4 <= img.size()) { const uint32_t lie = (rng() % 2) ? 0xFFFFFFF0u + (rng() % 16) : (uint32_t)rng(); std::memcpy(&img[at], &lie, 4); } break;   // lying size field
                case 2: img.resize(at); break;                                                                                                                         // truncation
                default: img.insert(img.begin() + (std::ptrdiff_t)at, (size_t)(rng() % 9), (uint8_t)0); break;
                }
            }
            // exact-size heap copy: any read past n_bytes is an ASan report
            std::vector<uint8_t> exact(img);
            exact.shrink_to_fit();
            dabgpu_wav_header h;
            const int st = dabgpu_wav_parse_header(exact.empty() ? nullptr : exact.data(), exact.size(), &h);
            CHECK(st == DABGPU_OK || st == DABGPU_ERR_INVALID_ARG, "wav status %d", st);
            if (st == DABGPU_OK) {
                accepted_wav++;
                CHECK(h.data_chunk_offset <= exact.size(), "data offset %llu beyond the %zu-byte image", (unsigned long long)h.data_chunk_offset, exact.size());
                CHECK(h.iq_format >= DABGPU_IQ_WAV_PCM8 && h.iq_format < DABGPU_IQ_NB_FORMATS && (h.total_channels == 1 || h.total_channels == 2), "accepted header fields");
                CHECK(dabgpu_iq_format_sample_bytes(h.iq_format) == 2u * (h.bits_per_sample / 8u), "sample bytes of format %d", h.iq_format);
            }
            (void)dabgpu_wav_parse_header(exact.data(), exact.size(), nullptr);
        }
        // ---- codeword descriptors ----
        {
            dabgpu_codeword d;
            std::memset(&d, 0, sizeof(d));
            const bool sane = rng() % 2 == 0;
            uint32_t steps = 0;
            for (int k = 0; k < 4; k++) {
                d.seg_pi[k] = sane ? (uint32_t)pick(1, 24) : (uint32_t)wild();
                d.seg_steps[k] = sane ? 8u * (uint32_t)pick(0, 100) : (uint32_t)wild();
                steps += d.seg_steps[k];
            }
            d.n_steps = (rng() % 4) ? steps + 6 : (uint32_t)wild();
            d.d_src = (rng() % 8) ? 0x1000 : 0; d.d_out = (rng() % 8) ? 0x2000 : 0;
            d.n_slots = (rng() % 3) ? 0 : (uint32_t)wild();
            d.newest_slot = (uint32_t)wild(); d.cifs_per_frame = (uint32_t)pick(0, 5); d.cif_stride = (uint32_t)wild(); d.frame_stride = (uint32_t)wild();
            d.flags = (uint32_t)(rng() % 16);
            const int st = dabgpu_host_validate_codeword(d, (size_t)it);
            CHECK(st == DABGPU_OK || st == DABGPU_ERR_INVALID_ARG, "validate_codeword status %d", st);
            if (st == DABGPU_OK) {
                accepted_cw++;
                CHECK(d.d_src && d.d_out && d.n_steps >= 1, "accepted a codeword without addresses / steps");
                if (!(d.flags & DABGPU_CW_DEPUNCTURED)) CHECK(d.n_steps == steps + 6 && (steps % 8) == 0, "accepted n_steps %u for %u segment steps", d.n_steps, steps);
                if (d.n_slots) CHECK(d.n_slots >= 16 && d.cifs_per_frame > 0 && d.newest_slot < d.n_slots, "accepted ring geometry");
            }
        }
        // ---- cost model, run-length rules, tables ----
        {
            const size_t n_cw = (size_t)(rng() % 5 == 0 ? 0 : rng() % 500000), n_groups = (n_cw + 63) / 64;
            const double steps = (double)pick(1, 5000);
            const int forced = pick(0, 3);
            const int m = dabgpu_host_choose_mapping(forced, (rng() % 7 == 0) ? 0.0 : (double)pick(1, 2048), n_cw, n_groups, (double)n_cw * steps, (double)n_groups * steps, steps, rng() % 2);
            CHECK(m >= DABGPU_VIT_MAP_WAVE && m <= DABGPU_VIT_MAP_OCTET, "mapping %d", m);
            if (forced != DABGPU_VIT_MAP_AUTO) CHECK(m == forced, "a forced mapping must come back unchanged");
            const size_t nf = (size_t)(rng() % 3 ? rng() % 3000 : rng());
            const int spb = dabgpu_host_small_batch_spb(nf ? nf : 1);
            CHECK(spb >= 3 && spb <= 25, "small-batch run length %d for %zu frames", spb, nf);
            const int b = dabgpu_host_spb_bucket(nf);
            CHECK(b >= 0 && b <= 40 && (nf <= 1 || ((size_t)1 << b) >= nf || b == 40), "bucket %d of %zu", b, nf);
            const int mode = pick(-3, 8);
            int geom[9];
            const int gs = dabgpu_get_ofdm_params(mode, geom);
            CHECK((gs == DABGPU_OK) == (mode >= 1 && mode <= 4), "get_ofdm_params(%d) = %d", mode, gs);
            if (gs == DABGPU_OK) {
                std::vector<float> prs(2 * (size_t)geom[3]);
                std::vector<int> map((size_t)geom[5]);
                CHECK(dabgpu_get_prs_fft_ref(mode, prs.data()) == DABGPU_OK && dabgpu_get_carrier_mapper(mode, map.data()) == DABGPU_OK, "tables of mode %d", mode);
                std::vector<char> seen((size_t)geom[5], 0);
                for (int v : map) { CHECK(v >= 0 && v < geom[5] && !seen[(size_t)v], "mapper of mode %d is not a permutation", mode); if (v >= 0 && v < geom[5]) seen[(size_t)v] = 1; }
            } else {
                float dummy[4]; int idummy[4];
                CHECK(dabgpu_get_prs_fft_ref(mode, dummy) != DABGPU_OK && dabgpu_get_carrier_mapper(mode, idummy) != DABGPU_OK, "tables of an invalid mode");
            }
            std::string name;
            for (unsigned k = rng() % 12; k > 0; k--) name.push_back((char)(rng() % 96 + 32));
            const int f = dabgpu_iq_format_from_mode((rng() % 4) ? name.c_str() : "raw_s16l");
            CHECK(f >= -1 && f < 14, "format %d", f);
            CHECK(dabgpu_iq_format_sample_bytes(wild()) <= 16, "sample bytes");
            (void)dabgpu_iq_format_from_mode(nullptr);
            (void)dabgpu_strerror(wild());
        }
    }
    // the Viterbi constant tables against an independent statement of ETSI EN 300 401 table 13 and clause 10
    {
        dabgpu_vit_tables T;
        dabgpu_host_fill_vit_tables(&T);
        for (int pi = 1; pi <= 24; pi++) {
            int total = 0;
            for (int g = 0; g < 8; g++) { const int cnt = T.pi_tab[pi * 8 + g] & 0xFF, pre = T.pi_tab[pi * 8 + g] >> 8; CHECK(cnt >= 1 && cnt <= 4 && pre == total, "PI_%d group %d", pi, g); total += cnt; }
            CHECK(total == 8 + pi, "PI_%d keeps %d of 32", pi, total);
        }
        unsigned reg = 0x1FF;                             // x^9 + x^5 + 1, all ones
        for (int k = 0; k < 64; k++) {
            unsigned byte = 0;
            for (int i = 0; i < 8; i++) { const unsigned v = ((reg >> 8) ^ (reg >> 4)) & 1u; byte = (byte << 1) | v; reg = ((reg << 1) | v) & 0x1FF; }
            CHECK(T.prbs[k] == byte, "energy-dispersal byte %d", k);
        }
    }
    std::printf("{\"iterations\": %d, \"accepted_plans\": %ld, \"accepted_wav\": %ld, \"accepted_codewords\": %ld, \"failed_checks\": %d}\n", iters, accepted_plans, accepted_wav,
                accepted_cw, g_fail);
    return g_fail ? 1 : 0;
}
