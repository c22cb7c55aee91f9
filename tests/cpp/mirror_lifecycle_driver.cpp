// Decoders that come and go while the stream runs -- the way basic_radio creates a channel's MSC_Decoder when the FIG database completes an entry,
// some frames into the stream, and drops it when the user deselects the service (src/basic_radio/basic_radio.cpp:67-120 UpdateAfterProcessing,
// basic_audio_channel.cpp:12).  The classes with the reference's signatures only: OFDM_Demod -> On_OFDM_Frame observer -> FIC_Decoder (optional,
// from a given frame on) + the MSC_Decoders alive at that frame.
//
//   mirror_lifecycle_driver <iq.c32> <out_dir> <block_size> <script.txt>
//   script lines:  <frame> add <id> <start_cu> <length_cu> <eep_level> <eep_type_b> [<is_uep> <uep_index>]      before frame <frame> is decoded
//                  <frame> del <id>
//                  <frame> fic <0|1>
// Output: out_dir/msc_<id>.bin = records { u32 frame, u32 cif, u32 n, n bytes } of every DecodeCIF call of decoder <id> (n = 0 while its time
// de-interleaver fills), out_dir/fibs.bin = records { u32 frame, 30 bytes } of every CRC-valid FIB; stdout: frames=<n> and how many FIB groups / CIFs
// the decoders picked up from the frames' batched decodes and how many they decoded call by call.
// The expected files are composed from the oracle in tests/test_mirror_lifecycle.py (CPU: linked with tests/cpp/fake_dabgpu_oracle.cpp;
// -m gpu: with libdabgpu.so, where every change of the set is a new decode layout of the receiver's frame session while frames are in flight).
#include <algorithm>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "dab/constants/dab_parameters.h"
#include "dab/dabgpu_frame_batcher.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/msc_decoder.h"
#include "ofdm/ofdm_helpers.h"

namespace {
void append(const std::string& path, const void* data, size_t n) {
    std::ofstream f(path, std::ios::binary | std::ios::app);
    f.write(static_cast<const char*>(data), (std::streamsize)n);
}
struct Event { int frame; std::string op; int id, start, length, level, type_b, is_uep, uep_index; };
}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s iq.c32 out_dir block_size script.txt\n", argv[0]); return 2; }
    const std::string out = argv[2];
    const size_t block = (size_t)std::atol(argv[3]);
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    std::vector<Event> script;
    {
        std::ifstream sf(argv[4]);
        std::string line;
        while (std::getline(sf, line)) {
            std::istringstream ls(line);
            Event e{0, "", 0, 0, 0, 0, 0, 0, 0};
            if (!(ls >> e.frame >> e.op)) continue;
            if (e.op == "add") { ls >> e.id >> e.start >> e.length >> e.level >> e.type_b; if (!(ls >> e.is_uep >> e.uep_index)) e.is_uep = e.uep_index = 0; }
            else ls >> e.id;
            script.push_back(e);
        }
    }
    const DAB_Parameters dab = get_dab_parameters(1);
    auto demod = Create_OFDM_Demodulator(1);
    std::unique_ptr<FIC_Decoder> fic;
    std::map<int, std::unique_ptr<MSC_Decoder>> msc;
    uint32_t frame = 0;
    demod->On_OFDM_Frame().Attach([&](tcb::span<const viterbi_bit_t> bits) {
        for (const Event& e : script) {
            if (e.frame != (int)frame) continue;
            if (e.op == "add") {
                Subchannel sc((subchannel_id_t)e.id);
                sc.start_address = (subchannel_addr_t)e.start;
                sc.length = (subchannel_size_t)e.length;
                sc.eep_prot_level = (eep_protection_level_t)e.level;
                sc.eep_type = e.type_b ? EEP_Type::TYPE_B : EEP_Type::TYPE_A;
                sc.is_uep = e.is_uep != 0;
                sc.uep_prot_index = (uep_protection_index_t)e.uep_index;
                sc.is_complete = true;
                msc[e.id] = std::make_unique<MSC_Decoder>(sc);
            } else if (e.op == "del") {
                msc.erase(e.id);
            } else if (e.op == "fic") {
                if (e.id && !fic) {
                    fic = std::make_unique<FIC_Decoder>((size_t)dab.nb_fib_cif_bits, (size_t)dab.nb_fibs_per_cif);
                    fic->OnFIB().Attach([&](tcb::span<const uint8_t> fib) { append(out + "/fibs.bin", &frame, 4); append(out + "/fibs.bin", fib.data(), fib.size()); });
                } else if (!e.id) {
                    fic.reset();
                }
            }
        }
        auto fic_bits = bits.subspan(0, (size_t)dab.nb_fic_bits);
        auto msc_bits = bits.subspan((size_t)dab.nb_fic_bits, (size_t)dab.nb_msc_bits);
        if (fic)
            for (int c = 0; c < dab.nb_cifs; c++) fic->DecodeFIBGroup(fic_bits.subspan((size_t)c * dab.nb_fib_cif_bits, (size_t)dab.nb_fib_cif_bits), (size_t)c);
        for (uint32_t c = 0; c < (uint32_t)dab.nb_cifs; c++) {
            auto cif = msc_bits.subspan((size_t)c * dab.nb_cif_bits, (size_t)dab.nb_cif_bits);
            for (auto& kv : msc) {
                auto got = kv.second->DecodeCIF(cif);
                const uint32_t n = (uint32_t)got.size();
                const std::string path = out + "/msc_" + std::to_string(kv.first) + ".bin";
                append(path, &frame, 4); append(path, &c, 4); append(path, &n, 4); append(path, got.data(), got.size());
            }
        }
        frame++;
    });
    // DABGPU_HARNESS_SCHEDULE as in mirror_harness.cpp: a text file of block lengths, one per Process() call; a negative entry = Reset() first
    std::vector<long> schedule;
    if (const char* sp = std::getenv("DABGPU_HARNESS_SCHEDULE")) {
        std::ifstream sf(sp);
        long v;
        while (sf >> v) if (v != 0) schedule.push_back(v);
    }
    size_t max_block = block;
    for (long v : schedule) max_block = std::max(max_block, (size_t)std::labs(v));
    std::vector<std::complex<float>> buf(max_block);
    for (size_t call = 0; in; call++) {
        long want = schedule.empty() ? (long)block : schedule[std::min(call, schedule.size() - 1)];
        if (want < 0) { demod->Reset(); want = -want; }
        in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)((size_t)want * sizeof(std::complex<float>)));
        const size_t got = (size_t)in.gcount() / sizeof(std::complex<float>);
        if (got == 0) break;
        demod->Process(tcb::span<const std::complex<float>>(buf.data(), got));
    }
    demod->Synchronize();
    const auto k = dabgpu_frame_batcher::counters();
    std::printf("frames=%u fib_groups_batched=%llu fib_groups_call_by_call=%llu cifs_batched=%llu cifs_call_by_call=%llu\n", frame, k.fib_groups_batched,
                k.fib_groups_call_by_call, k.cifs_batched, k.cifs_call_by_call);
    return 0;
}
