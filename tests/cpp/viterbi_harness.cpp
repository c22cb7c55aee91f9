// viterbi_harness.cpp -- drives the DAB_Viterbi_Decoder mirror class through scripted sequences of reset / update / chainback calls
// (any puncture vector, any lengths, any chunking, any start / end state) and writes what every call returned;
// tests/test_gpu_viterbi_general.py runs the same script through the oracle's Viterbi class and compares byte for byte.
//
//   viterbi_harness <script.bin> <out.bin>
// script: u32 n_cases; per case: u32 start_state, end_state, n_out_bytes, n_updates;
//         per update: u32 n_code, u8 code[n_code], u32 requested, u32 n_punctured, i8 punctured[n_punctured]
// out:    per case: per update u64 consumed; u64 current_decoded_bit; u32 status (0 ok, 1 std::invalid_argument, 2 other exception);
//         u64 path_error; u8 bytes[n_out_bytes]
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <vector>

#include "dab/algorithms/dab_viterbi_decoder.h"

template <typename T> static bool rd(std::ifstream& f, T& v) { return (bool)f.read(reinterpret_cast<char*>(&v), sizeof(T)); }
template <typename T> static void wr(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof(T)); }

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s script.bin out.bin\n", argv[0]); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    if (!in || !out) { std::fprintf(stderr, "cannot open files\n"); return 2; }
    uint32_t n_cases = 0;
    if (!rd(in, n_cases)) return 2;
    DAB_Viterbi_Decoder vit;                         // ONE object for all cases: reset() must leave nothing behind
    vit.set_traceback_length(1 << 16);
    for (uint32_t c = 0; c < n_cases; c++) {
        uint32_t start = 0, end = 0, n_out = 0, n_upd = 0;
        if (!rd(in, start) || !rd(in, end) || !rd(in, n_out) || !rd(in, n_upd)) return 2;
        vit.reset(start);
        for (uint32_t u = 0; u < n_upd; u++) {
            uint32_t n_code = 0, requested = 0, n_p = 0;
            if (!rd(in, n_code)) return 2;
            std::vector<uint8_t> code(n_code);
            in.read(reinterpret_cast<char*>(code.data()), n_code);
            if (!rd(in, requested) || !rd(in, n_p)) return 2;
            std::vector<viterbi_bit_t> p(n_p);
            in.read(reinterpret_cast<char*>(p.data()), n_p);
            const uint64_t used = vit.update(p, code, requested);
            wr(out, used);
        }
        wr(out, (uint64_t)vit.get_current_decoded_bit());
        std::vector<uint8_t> bytes(n_out, 0xEE);
        uint32_t status = 0;
        uint64_t err = 0;
        try { err = vit.chainback(bytes, end); }
        catch (const std::invalid_argument&) { status = 1; }
        catch (const std::exception& e) { status = 2; std::fprintf(stderr, "case %u: %s\n", c, e.what()); }
        wr(out, status); wr(out, err);
        out.write(reinterpret_cast<const char*>(bytes.data()), n_out);
    }
    return 0;
}
