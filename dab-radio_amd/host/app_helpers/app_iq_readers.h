// app_helpers/app_iq_readers.h -- IQ file readers with the reference's mode strings and sample arithmetic
// (examples/app_helpers/app_iq_readers.h:107-159), dequantised on the device: the raw bytes of a block go to
// libdabgpu.so (dabgpu_iq_convert_host_sync) and come back as std::complex<float>.
//
//   auto reader = get_iq_file_reader_from_mode_string(fp, "raw_u8");     // throws std::runtime_error like the reference
//   size_t n = reader->read(block);                                      // whole IQ samples; a trailing partial sample is dropped
//
// Differences from the reference, on purpose: takes the FILE* directly (the reference wraps it in InputFile<uint8_t>),
// and a wav payload ends at the data chunk's declared size (the reference's reader never advances its byte counter,
// app_wav_reader.h:470-479, and so also returns whatever follows the data chunk).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <complex>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "./app_io_buffers.h"
#include "dab/dabgpu_shared_context.h"
#include "dabgpu.h"

static const std::vector<std::string> iq_read_modes = {
    "wav",
    "raw_u8", "raw_s8",
    "raw_s16l", "raw_s16b", "raw_u16l", "raw_u16b",
    "raw_s32l", "raw_s32b", "raw_u32l", "raw_u32b",
    "raw_f32l", "raw_f32b", "raw_f64l", "raw_f64b",
};

class DeviceIQFileReader: public InputBuffer<std::complex<float>>
{
private:
    FILE* m_file = nullptr;
    int m_format = -1;
    size_t m_sample_bytes = 0;
    std::vector<uint8_t> m_prefix;          // bytes already read past the wav header
    size_t m_prefix_pos = 0;
    uint64_t m_remaining = UINT64_MAX;      // payload bytes left (wav data chunk size)
    std::vector<uint8_t> m_raw;
    dabgpu_wav_header m_wav_header{};
    bool m_is_wav = false;
public:
    DeviceIQFileReader(FILE* file, const std::string& mode): m_file(file) {
        if (file == nullptr) throw std::runtime_error("IQ reader: null file");
        if (mode == "wav") {
            m_is_wav = true;
            // the header walk needs every chunk up to "data": read a prefix, retry with a larger one if it ends early
            size_t want = 4096;
            for (;;) {
                const size_t have = m_prefix.size();
                m_prefix.resize(want);
                const size_t got = fread(m_prefix.data() + have, 1, want - have, m_file);
                m_prefix.resize(have + got);
                const int st = dabgpu_wav_parse_header(m_prefix.data(), m_prefix.size(), &m_wav_header);
                if (st == DABGPU_OK) break;
                if (got == 0 || want >= (size_t(1) << 28)) throw std::runtime_error(std::string("wav: ") + dabgpu_last_error());
                want *= 4;
            }
            if (m_wav_header.total_channels != 2) {
                throw std::runtime_error("WAV file should have 2 channels for IQ stream but got " +
                                         std::to_string(m_wav_header.total_channels) + " channels");
            }
            m_format = m_wav_header.iq_format;
            m_prefix_pos = size_t(m_wav_header.data_chunk_offset);
            m_remaining = m_wav_header.data_chunk_size;
        } else {
            m_format = dabgpu_iq_format_from_mode(mode.c_str());
            if (m_format < 0) throw std::runtime_error("Unknown iq file format: '" + mode + "'");
        }
        m_sample_bytes = dabgpu_iq_format_sample_bytes(m_format);
    }
    ~DeviceIQFileReader() override = default;
    int get_format() const { return m_format; }
    bool is_wav() const { return m_is_wav; }
    const dabgpu_wav_header& get_wav_header() const { return m_wav_header; }
    // raw payload bytes of up to n whole samples; returns the number of whole samples (the raw-capture path of the CLI)
    size_t read_raw(size_t n, std::vector<uint8_t>& raw) {
        uint64_t want = uint64_t(n) * m_sample_bytes;
        if (want > m_remaining) want = m_remaining;
        raw.resize(size_t(want));
        size_t got = 0;
        if (m_prefix_pos < m_prefix.size()) {
            const size_t k = std::min(m_prefix.size() - m_prefix_pos, size_t(want));
            std::copy_n(m_prefix.data() + m_prefix_pos, k, raw.data());
            m_prefix_pos += k;
            got = k;
        }
        if (got < want) got += fread(raw.data() + got, 1, size_t(want) - got, m_file);
        if (m_remaining != UINT64_MAX) m_remaining -= got;
        const size_t n_samples = got / m_sample_bytes;
        raw.resize(n_samples * m_sample_bytes);
        return n_samples;
    }
    size_t read(tcb::span<std::complex<float>> dest) override {
        const size_t n = read_raw(dest.size(), m_raw);
        if (n == 0) return 0;
        const int st = dabgpu_iq_convert_host_sync(dabgpu_shared_context(), m_raw.data(), m_format, n,
                                                   reinterpret_cast<float*>(dest.data()));
        if (st != DABGPU_OK) throw std::runtime_error(std::string("dabgpu_iq_convert_host_sync: ") + dabgpu_last_error());
        return n;
    }
};

static inline std::shared_ptr<DeviceIQFileReader> get_iq_file_reader_from_mode_string(FILE* file, const std::string& mode) {
    return std::make_shared<DeviceIQFileReader>(file, mode);
}
