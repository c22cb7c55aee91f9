// dabplus_harness.cpp -- drives the AAC_Frame_Processor mirror class the way Basic_DAB_Plus_Channel does
// (src/basic_radio/basic_dab_plus_channel.cpp:47-60): one Process() per logical frame, every callback logged as a line
// of integers that tests/test_gpu_dabplus_mirror.py compares with the reference's own callbacks (golden vectors).
//   dabplus_harness <frames.bin> <frame_bytes>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "dab/audio/aac_frame_processor.h"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s frames.bin frame_bytes\n", argv[0]); return 2; }
    const size_t n = (size_t)std::atol(argv[2]);
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) return 2;
    AAC_Frame_Processor proc;
    int k = 0;
    proc.OnFirecodeError().Attach([&](int idx, uint16_t got, uint16_t calc) { std::printf("%d firecode %d %u %u\n", k, idx, got, calc); });
    proc.OnRSError().Attach([&](int i, int total) { std::printf("%d rs %d %d\n", k, i, total); });
    proc.OnSuperFrameHeader().Attach([&](SuperFrameHeader h) {
        std::printf("%d header %u %d %d %d %d\n", k, h.sampling_rate, (int)h.is_parametric_stereo, (int)h.is_spectral_band_replication,
                    (int)h.is_stereo, (int)h.mpeg_surround);
    });
    proc.OnAccessUnitCRCError().Attach([&](int i, int total, uint16_t got, uint16_t calc) { std::printf("%d aucrc %d %d %u %u\n", k, i, total, got, calc); });
    proc.OnAccessUnit().Attach([&](int i, int total, tcb::span<uint8_t> d) {
        unsigned sum = 0;
        for (uint8_t b : d) sum = sum * 31u + b;
        std::printf("%d au %d %d %zu %u\n", k, i, total, d.size(), sum);
    });
    std::vector<uint8_t> frame(n);
    while (in.read(reinterpret_cast<char*>(frame.data()), (std::streamsize)n)) {
        proc.Process(frame);
        k++;
    }
    return 0;
}
