// ref_callers_driver.cpp -- TEST INFRASTRUCTURE: the REFERENCE's own callers of the hot path, executed over the mirror classes.
// It only compiles inside an overlay of /root/reference (tests/test_reference_callers_run.py builds it; the GPU box has no reference):
//   * examples/app_helpers/app_iq_readers.h  get_iq_file_reader_from_mode_string("raw_u8")        the reference's capture reader
//   * examples/app_helpers/app_ofdm_blocks.h:25-58  OFDM_Block (constructor from the reference's PRS / mapper / parameter objects,
//     On_OFDM_Frame().Attach, run(block_size)) on its own thread, writing into the reference's ThreadedRingBuffer exactly as
//     examples/basic_radio_app.cpp:262-322,406-416 wires the two blocks together
//   * src/basic_radio/basic_fic_runner.cpp:20-49  BasicFICRunner (mirror FIC_Decoder -> the reference's FIG_Processor ->
//     Radio_FIG_Handler -> DAB_Database_Updater)
//   * src/basic_radio/basic_radio.cpp:41-65 Process and :83-154 UpdateAfterProcessing: BasicRadio itself cannot be linked here (its audio
//     channel classes pull in vendor/faad2 and vendor/mpg123, empty submodules, and no stand-ins are written), so `Radio` below
//     restates those two functions statement by statement over the reference's own database objects; where the reference makes a
//     Basic_DAB_Plus_Channel / Basic_DAB_Channel / Basic_Data_Packet_Channel -- each of which does make_unique<MSC_Decoder>(subchannel)
//     (basic_audio_channel.cpp:12, basic_data_packet_channel.cpp:26) and calls DecodeCIF per CIF (basic_dab_plus_channel.cpp:47-51,
//     basic_dab_channel.cpp:53, basic_data_packet_channel.cpp:60) -- it makes the MSC_Decoder and calls DecodeCIF.
// Everything below the class boundary is whatever dabgpu.h implementation the executable is linked with: the oracle-backed one
// (tests/cpp/fake_dabgpu_oracle.cpp) in the build container.
//
//   ref_callers_driver <capture.raw_u8> <out_dir> <block_size>
// writes out_dir/frame_bits.bin (the OFDM_Block's output stream), fibs.bin {u32 frame, 30 bytes} for every CRC-valid FIB (a second
// FIC_Decoder beside the runner's: BasicFICRunner keeps its own private), created.txt "<frame> <subchannel id> <kind> <start> <length>
// <is_uep> <uep_index> <eep_level> <eep_type_b> <fec>": the decoder exists from the frame AFTER <frame> on, msc_<id>.bin
// {u32 frame, u32 cif, u32 n, n bytes} per DecodeCIF call, database.txt (sorted, canonical) after the last frame.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <complex>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <fmt/format.h>
#include "app_helpers/app_io_buffers.h"
#include "app_helpers/app_iq_readers.h"
#include "app_helpers/app_ofdm_blocks.h"
#include "basic_radio/basic_fic_runner.h"
#include "dab/constants/dab_parameters.h"
#include "dab/database/dab_database.h"
#include "dab/database/dab_database_entities.h"
#include "dab/database/dab_database_updater.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/msc_decoder.h"

namespace {
void append(const std::string& path, const void* data, size_t n) {
    std::ofstream f(path, std::ios::binary | std::ios::app);
    f.write(static_cast<const char*>(data), (std::streamsize)n);
}

struct Runner {                                   // what the reference's three channel classes have in common on this path
    Subchannel subchannel;
    std::unique_ptr<MSC_Decoder> decoder;
    std::string kind;
};

class Radio {                                     // basic_radio.cpp:25-65,83-154 without the audio / data payload decoders
public:
    const DAB_Parameters m_params;
    std::unique_ptr<BasicFICRunner> m_fic_runner;
    std::unique_ptr<FIC_Decoder> m_fib_tap;
    DAB_Database m_dab_database;
    DatabaseUpdaterGlobalStatistics m_dab_database_stats;
    std::map<subchannel_id_t, std::shared_ptr<Runner>> m_msc_runners;
    std::string m_out;
    std::atomic<uint32_t> m_frame{0};

    Radio(const DAB_Parameters& params, const std::string& out) : m_params(params), m_out(out) {
        m_fic_runner = std::make_unique<BasicFICRunner>(m_params);
        m_fib_tap = std::make_unique<FIC_Decoder>((size_t)m_params.nb_fib_cif_bits, (size_t)m_params.nb_fibs_per_cif);
        m_fib_tap->OnFIB().Attach([this](tcb::span<const uint8_t> fib) { const uint32_t fr = m_frame; append(m_out + "/fibs.bin", &fr, 4); append(m_out + "/fibs.bin", fib.data(), fib.size()); });
    }

    void Process(tcb::span<const viterbi_bit_t> buf) {            // :41-65
        const int N = (int)buf.size();
        if (N != m_params.nb_frame_bits) return;
        auto fic_buf = buf.subspan(0, (size_t)m_params.nb_fic_bits);
        auto msc_buf = buf.subspan((size_t)m_params.nb_fic_bits, (size_t)m_params.nb_msc_bits);
        m_fic_runner->Process(fic_buf);
        for (int i = 0; i < m_params.nb_cifs; i++)
            m_fib_tap->DecodeFIBGroup(fic_buf.subspan((size_t)(i * m_params.nb_fib_cif_bits), (size_t)m_params.nb_fib_cif_bits), (size_t)i);
        const uint32_t frame = m_frame;
        for (const auto& [id, runner] : m_msc_runners) {
            for (uint32_t i = 0; i < (uint32_t)m_params.nb_cifs; i++) {                           // basic_dab_plus_channel.cpp:47-51
                const auto cif_buf = msc_buf.subspan((size_t)i * (size_t)m_params.nb_cif_bits, (size_t)m_params.nb_cif_bits);
                const auto decoded_bytes = runner->decoder->DecodeCIF(cif_buf);
                const uint32_t n = (uint32_t)decoded_bytes.size();
                const std::string path = m_out + "/msc_" + std::to_string((int)id) + ".bin";
                append(path, &frame, 4); append(path, &i, 4); append(path, &n, 4); append(path, decoded_bytes.data(), decoded_bytes.size());
            }
        }
        UpdateAfterProcessing();
        m_frame++;
    }

    void UpdateAfterProcessing() {                                  // :83-154
        const auto& dab_database_updater = m_fic_runner->GetDatabaseUpdater();
        const auto& new_dab_database = dab_database_updater.GetDatabase();
        const auto& new_dab_database_stats = dab_database_updater.GetStatistics();
        const bool is_updated = new_dab_database_stats != m_dab_database_stats;
        if (!is_updated) return;
        m_dab_database = new_dab_database;
        m_dab_database_stats = new_dab_database_stats;
        for (auto& subchannel : m_dab_database.subchannels) {
            if (!subchannel.is_complete) continue;
            if (m_msc_runners.find(subchannel.id) != m_msc_runners.end()) continue;
            const ServiceComponent* service_component = nullptr;
            for (auto& e : m_dab_database.service_components) {
                if (e.subchannel_id == subchannel.id) { service_component = &e; break; }
            }
            if (!service_component) continue;
            if (!service_component->is_complete) continue;
            const auto mode = service_component->transport_mode;
            const auto audio_type = service_component->audio_service_type;
            if (audio_type == AudioServiceType::DAB_PLUS && mode == TransportMode::STREAM_MODE_AUDIO) { Add(subchannel, "dab_plus"); continue; }
            if (audio_type == AudioServiceType::DAB && mode == TransportMode::STREAM_MODE_AUDIO) { Add(subchannel, "dab"); continue; }
            if (mode == TransportMode::PACKET_MODE_DATA && (subchannel.fec_scheme != FEC_Scheme::UNDEFINED)) { Add(subchannel, "data_packet"); continue; }
        }
    }

    void Add(const Subchannel& subchannel, const char* kind) {
        auto runner = std::make_shared<Runner>(Runner{subchannel, std::make_unique<MSC_Decoder>(subchannel), kind});      // basic_audio_channel.cpp:12
        m_msc_runners.insert({subchannel.id, runner});
        const std::string line = fmt::format("{} {} {} {} {} {} {} {} {} {}\n", m_frame.load(), (int)subchannel.id, kind, (int)subchannel.start_address, (int)subchannel.length,
                                             (int)subchannel.is_uep, (int)subchannel.uep_prot_index, (int)subchannel.eep_prot_level,
                                             subchannel.eep_type == EEP_Type::TYPE_B ? 1 : 0, (int)subchannel.fec_scheme);
        append(m_out + "/created.txt", line.data(), line.size());
    }

    void DumpDatabase() {
        const DAB_Database& db = m_fic_runner->GetDatabaseUpdater().GetDatabase();
        std::vector<std::string> subs, services, comps;
        const auto& e = db.ensemble;
        std::string text = fmt::format("ensemble id={:04X} ecc={:02X} lto={} inter_table={} label=[{}]\n", e.id.value, e.extended_country_code, (int)e.local_time_offset,
                                       (int)e.international_table_id, e.label);
        auto fec = [](FEC_Scheme f) { return f == FEC_Scheme::UNDEFINED ? std::string("none") : std::to_string((int)f); };
        std::vector<Subchannel> ss = db.subchannels;
        std::sort(ss.begin(), ss.end(), [](const Subchannel& a, const Subchannel& b) { return a.id < b.id; });
        for (const auto& s : ss) {
            if (s.is_uep) text += fmt::format("subchannel id={} start={} length={} uep index={} fec={} complete={}\n", (int)s.id, (int)s.start_address, (int)s.length,
                                              (int)s.uep_prot_index, fec(s.fec_scheme), (int)s.is_complete);
            else text += fmt::format("subchannel id={} start={} length={} eep level={} type={} fec={} complete={}\n", (int)s.id, (int)s.start_address, (int)s.length,
                                     (int)s.eep_prot_level, s.eep_type == EEP_Type::TYPE_A ? "A" : (s.eep_type == EEP_Type::TYPE_B ? "B" : "?"), fec(s.fec_scheme), (int)s.is_complete);
        }
        std::vector<Service> sv = db.services;
        std::sort(sv.begin(), sv.end(), [](const Service& a, const Service& b) { return a.id.value < b.id.value; });
        for (const auto& s : sv)
            text += fmt::format("service id={:X} bits={} label=[{}] pty={}\n", s.id.value, s.id.type == ServiceIdType::BITS32 ? 32 : (s.id.type == ServiceIdType::BITS16 ? 16 : 24),
                                s.label, (int)s.programme_type);
        std::vector<ServiceComponent> sc = db.service_components;
        std::sort(sc.begin(), sc.end(), [](const ServiceComponent& a, const ServiceComponent& b) { return a.service_id.value < b.service_id.value; });
        for (const auto& c : sc) {
            text += fmt::format("component service={:X} scids={} subchannel={} ", c.service_id.value, (int)c.component_id, (int)c.subchannel_id);
            if (c.transport_mode == TransportMode::STREAM_MODE_AUDIO) text += fmt::format("mode=stream_audio audio={} complete={}\n", (int)c.audio_service_type, (int)c.is_complete);
            else if (c.transport_mode == TransportMode::STREAM_MODE_DATA) text += fmt::format("mode=stream_data data={} complete={}\n", (int)c.data_service_type, (int)c.is_complete);
            else if (c.transport_mode == TransportMode::PACKET_MODE_DATA) {
                int apps = 0;
                for (auto a : c.application_types) apps += (int)a;
                text += fmt::format("mode=packet_data data={} scid={} packet_addr={} apps={} complete={}\n", (int)c.data_service_type, (int)c.global_id, (int)c.packet_address, apps, (int)c.is_complete);
            } else text += "mode=undefined\n";
        }
        append(m_out + "/database.txt", text.data(), text.size());
    }
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s capture.raw_u8 out_dir block_size\n", argv[0]); return 2; }
    const std::string out = argv[2];
    const size_t block_size = (size_t)std::atol(argv[3]);
    FILE* fp_in = std::fopen(argv[1], "rb");
    if (!fp_in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    const int transmission_mode = 1;
    const auto dab_params = get_dab_parameters(transmission_mode);
    // examples/basic_radio_app.cpp:262-322: file -> IQ reader -> OFDM_Block -> splitter -> {ring buffer -> radio, soft-bit file}
    auto ofdm_block = std::make_shared<OFDM_Block>(transmission_mode, 1);
    auto ofdm_output_splitter = std::make_shared<OutputSplitter<viterbi_bit_t>>();
    ofdm_block->set_output_stream(ofdm_output_splitter);
    auto raw_iq_in = std::make_shared<InputFile<uint8_t>>(fp_in);
    auto iq_stream = get_iq_file_reader_from_mode_string(raw_iq_in, "raw_u8");
    ofdm_block->set_input_stream(iq_stream);
    FILE* fp_bits = std::fopen((out + "/frame_bits.bin").c_str(), "wb");
    auto soft_bits_out = std::make_shared<OutputFile<viterbi_bit_t>>(fp_bits);
    ofdm_output_splitter->add_output_stream(soft_bits_out);
    auto ofdm_to_radio_buffer = std::make_shared<ThreadedRingBuffer<viterbi_bit_t>>((size_t)dab_params.nb_frame_bits * 2);
    ofdm_output_splitter->add_output_stream(ofdm_to_radio_buffer);

    Radio radio(dab_params, out);
    std::thread thread_ofdm([&]() {                                  // :407-413
        ofdm_block->run(block_size);
#ifdef DABGPU_MIRROR_HAS_SYNCHRONIZE
        ofdm_block->get_ofdm_demod().Synchronize();                  // the mirror delivers frames from its own thread: drain before the stream is closed
#endif
        // ThreadedRingBuffer::read gives up a half-read frame when the buffer is closed under it (app_io_buffers.h:224-238; the reference's app
        // loses its last frame that way): let the radio side finish what the demodulator delivered before closing
        const uint32_t delivered = (uint32_t)ofdm_block->get_ofdm_demod().GetTotalFramesRead();
        while (radio.m_frame.load() < delivered) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        ofdm_to_radio_buffer->close();
    });
    std::thread thread_radio([&]() {                                 // app_radio_blocks.h:30-37
        std::vector<viterbi_bit_t> bits((size_t)dab_params.nb_frame_bits);
        while (true) {
            const size_t length = ofdm_to_radio_buffer->read(bits);
            if (length != bits.size()) return;
            radio.Process(bits);
        }
    });
    thread_ofdm.join();
    thread_radio.join();
    soft_bits_out->close();
    radio.DumpDatabase();
    const auto& d = ofdm_block->get_ofdm_demod();
    std::printf("frames=%u read=%d desync=%d decoders=%zu\n", radio.m_frame.load(), d.GetTotalFramesRead(), d.GetTotalFramesDesync(), radio.m_msc_runners.size());
    return 0;
}
