// ref_harness_dabplus.cpp -- C entry points around the reference's OWN DAB+ outer-code objects
// (src/dab/audio/aac_frame_processor.cpp, src/dab/algorithms/reed_solomon_decoder.cpp, compiled in place by
// oracle/Makefile into oracle/_ref/libdab_ref.so).  TEST INFRASTRUCTURE: pins oracle/dab_oracle_dabplus.c.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "dab/algorithms/reed_solomon_decoder.h"
#include "dab/audio/aac_frame_processor.h"

struct RefAac {
    AAC_Frame_Processor proc;
    // events of the current Process() call
    int firecode_errors = 0;
    int rs_failed_index = -1;
    int header_valid = 0;
    SuperFrameHeader header;
    int num_aus = 0;
    uint32_t au_ok_mask = 0, au_bad_mask = 0;
    std::vector<uint8_t> au_data[6];
    // the same callbacks as text lines (format of tests/cpp/dabplus_harness.cpp), k = index of the Process() call
    std::string log;
    int k = 0;
    void line(const char* fmt, long a = 0, long b = 0, long c = 0, long d = 0, long e = 0) {
        char buf[160];
        snprintf(buf, sizeof(buf), fmt, k, a, b, c, d, e);
        log += buf;
    }
    RefAac() {
        proc.OnFirecodeError().Attach([this](int idx, uint16_t got, uint16_t calc) { line("%d firecode %ld %ld %ld\n", idx, got, calc); });
        proc.OnRSError().Attach([this](int i, int total) { line("%d rs %ld %ld\n", i, total); });
        proc.OnSuperFrameHeader().Attach([this](SuperFrameHeader h) {
            line("%d header %ld %ld %ld %ld %ld\n", h.sampling_rate, h.is_parametric_stereo, h.is_spectral_band_replication, h.is_stereo, (long)h.mpeg_surround);
        });
        proc.OnAccessUnitCRCError().Attach([this](int i, int total, uint16_t got, uint16_t calc) { line("%d aucrc %ld %ld %ld %ld\n", i, total, got, calc); });
        proc.OnAccessUnit().Attach([this](int i, int total, tcb::span<uint8_t> d) {
            unsigned sum = 0;
            for (uint8_t b : d) sum = sum * 31u + b;
            line("%d au %ld %ld %ld %ld\n", i, total, (long)d.size(), (long)sum);
        });
        proc.OnFirecodeError().Attach([this](int, uint16_t, uint16_t) { firecode_errors++; });
        proc.OnRSError().Attach([this](int i, int) { rs_failed_index = i; });
        proc.OnSuperFrameHeader().Attach([this](SuperFrameHeader h) { header_valid = 1; header = h; });
        proc.OnAccessUnitCRCError().Attach([this](int i, int n, uint16_t, uint16_t) { num_aus = n; au_bad_mask |= 1u << i; });
        proc.OnAccessUnit().Attach([this](int i, int n, tcb::span<uint8_t> d) {
            num_aus = n; au_ok_mask |= 1u << i; au_data[i].assign(d.begin(), d.end());
        });
    }
};

extern "C" {

int ref_rs120_decode(uint8_t* cw, int* positions) {
    static Reed_Solomon_Decoder dec(8, 0b100011101, 0, 1, 10, 255 - 120);
    return dec.Decode(cw, positions, 0);
}

void* ref_aac_create() { return new RefAac(); }
void ref_aac_destroy(void* h) { delete static_cast<RefAac*>(h); }

// out12 = {firecode_errors, rs_failed_index, header_valid, sampling_rate, ps, sbr, stereo, surround, num_aus, au_ok_mask, au_bad_mask, 0}
// au_len[6], au_bytes[6][au_cap]
void ref_aac_process(void* h, const uint8_t* frame, int n, int32_t* out12, int32_t* au_len, uint8_t* au_bytes, int au_cap) {
    RefAac* r = static_cast<RefAac*>(h);
    r->firecode_errors = 0; r->rs_failed_index = -1; r->header_valid = 0; r->num_aus = 0; r->au_ok_mask = 0; r->au_bad_mask = 0;
    for (auto& v : r->au_data) v.clear();
    r->proc.Process({frame, (size_t)n});
    r->k++;
    out12[0] = r->firecode_errors; out12[1] = r->rs_failed_index; out12[2] = r->header_valid;
    out12[3] = (int32_t)r->header.sampling_rate; out12[4] = r->header.is_parametric_stereo; out12[5] = r->header.is_spectral_band_replication;
    out12[6] = r->header.is_stereo; out12[7] = (int32_t)r->header.mpeg_surround; out12[8] = r->num_aus;
    out12[9] = (int32_t)r->au_ok_mask; out12[10] = (int32_t)r->au_bad_mask; out12[11] = 0;
    for (int i = 0; i < 6; i++) {
        au_len[i] = (int32_t)r->au_data[i].size();
        if (!r->au_data[i].empty()) memcpy(au_bytes + (size_t)i * au_cap, r->au_data[i].data(), r->au_data[i].size() < (size_t)au_cap ? r->au_data[i].size() : (size_t)au_cap);
    }
}

// the callback log accumulated so far (NUL terminated), returns its length
long ref_aac_log(void* h, char* out, long cap) {
    RefAac* r = static_cast<RefAac*>(h);
    const long n = (long)r->log.size();
    if (out && cap > n) memcpy(out, r->log.c_str(), (size_t)n + 1);
    return n;
}

}  // extern "C"
