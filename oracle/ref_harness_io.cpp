// ref_harness_io.cpp -- C entry points around the reference's OWN file-format helpers, included from
// /root/reference where they lie (header-only: examples/app_helpers/app_iq_readers.h, app_wav_reader.h,
// app_viterbi_convert_block.h).  Built by oracle/Makefile into oracle/_ref/libdab_ref.so (never committed).
// TEST INFRASTRUCTURE: used to pin oracle/dab_oracle_io.c and to generate tests/golden vectors.
// fmt (the reference's vendor/fmt submodule is not vendored) comes from the copy PyTorch installs, header-only.
#include <stdint.h>
#include <stdio.h>
#include <complex>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "app_helpers/app_iq_readers.h"
#include "app_helpers/app_viterbi_convert_block.h"

extern "C" {

// the reference's reader chain for `mode` over the file at `path`, read in blocks of `block` samples
// returns the number of IQ samples produced, or -1 if the reference threw
long ref_iq_read_file(const char* path, const char* mode, float* out, size_t max_samples, size_t block) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return -2;
    try {
        auto file = std::make_shared<InputFile<uint8_t>>(fp);
        auto reader = get_iq_file_reader_from_mode_string(file, mode);
        auto* dst = reinterpret_cast<std::complex<float>*>(out);
        size_t total = 0;
        while (total < max_samples) {
            const size_t want = (max_samples - total < block) ? (max_samples - total) : block;
            const size_t got = reader->read(tcb::span<std::complex<float>>(dst + total, want));
            total += got;
            if (got != want) break;
        }
        return (long)total;
    } catch (const std::exception& e) {
        return -1;
    }
}

// wav_read_header + WavFileReader's constructor; out7 = {-, audio format code, channels, rate, bits, data size, data offset}
int ref_wav_header(const char* path, uint64_t* out7) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return -2;
    try {
        auto file = std::make_shared<InputFile<uint8_t>>(fp);
        WavFileReader reader(file);
        const auto& h = reader.get_header();
        out7[0] = 0;
        out7[1] = (uint64_t)static_cast<uint16_t>(h.format);
        out7[2] = h.total_channels;
        out7[3] = h.samples_per_second;
        out7[4] = h.bits_per_sample;
        out7[5] = h.data_chunk_size;
        out7[6] = h.data_chunk_offset;
        return 0;
    } catch (const std::exception& e) {
        return -1;
    }
}

void ref_bytes_to_bits(const uint8_t* bytes, size_t n_bytes, int8_t* bits) {
    convert_viterbi_bytes_to_bits({bytes, n_bytes}, {bits, n_bytes * 8});
}
void ref_bits_to_bytes(const int8_t* bits, size_t n_bytes, uint8_t* bytes) {
    convert_viterbi_bits_to_bytes({bits, n_bytes * 8}, {bytes, n_bytes});
}

}  // extern "C"
