// dabgpu_radio_cli.cpp -- command-line front end of the MI355X hot path with the option names of the reference's
// `basic_radio_app_cli` (examples/basic_radio_app.cpp:48-140) for the stages this library owns:
//
//   --configuration ofdm      IQ capture file  -> OFDM_Demod -> frame soft bits (or packed hard bytes) file
//   --configuration dab       frame bit file   -> FIC_Decoder / MSC_Decoder -> FIB bytes / sub-channel bytes files
//   --configuration dab+ofdm  both, bits handed over in memory
//
// The reference's application layer above the channel decoder (FIG parsing, database, audio, scraper, GUI) is not
// part of this library; where basic_radio_app hands the frame bits to BasicRadio, this tool writes what
// BasicRadio's first stage produces (basic_radio.cpp:41-65): the CRC-checked FIBs and, for sub-channels given
// explicitly with --radio-subchannel, the decoded logical frames.
//
// Output files are byte-for-byte what the reference writes with the same options (frame bits: 230400 int8 per frame,
// or 28800 bytes per frame with --ofdm-output-hard-bytes, LSB first).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <complex>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "app_helpers/app_iq_readers.h"
#include "app_helpers/app_viterbi_convert_block.h"
#include "dab/constants/dab_parameters.h"
#include "dab/fic/fic_decoder.h"
#include "dab/msc/msc_decoder.h"
#include "ofdm/ofdm_helpers.h"

struct Args {
    std::string input_file;
    int transmission_mode = 1;
    bool is_ofdm_used = true;
    bool is_dab_used = true;
    std::string ofdm_input_mode = "raw_u8";
    size_t ofdm_block_size = 65536;
    bool ofdm_disable_coarse_freq = false;
    bool ofdm_enable_output = false;
    std::string ofdm_output;
    bool ofdm_output_hard_bytes = false;
    bool radio_input_hard_bytes = false;
    std::string radio_fib_output;
    std::string radio_msc_output = "subchannel_";
    std::vector<Subchannel> subchannels;
};

static void usage(const char* argv0) {
    fprintf(stderr,
        "usage: %s [-i FILE] [--transmission-mode 1] [--configuration dab+ofdm|ofdm|dab]\n"
        "  [--ofdm-input-mode MODE] [--ofdm-block-size SAMPLES] [--ofdm-disable-coarse-freq]\n"
        "  [--ofdm-enable-output] [--ofdm-output FILE] [--ofdm-output-hard-bytes]\n"
        "  [--radio-input-hard-bytes] [--radio-fib-output FILE]\n"
        "  [--radio-subchannel START,LENGTH,EEP_LEVEL(1-4),EEP_TYPE(A|B) | START,LENGTH,uep,UEP_INDEX]...\n"
        "  [--radio-msc-output PREFIX]\n"
        "MODE: ", argv0);
    for (const auto& m : iq_read_modes) fprintf(stderr, "%s ", m.c_str());
    fprintf(stderr, "\n");
}

static Subchannel parse_subchannel(const std::string& text, size_t id) {
    std::vector<std::string> f;
    size_t a = 0;
    for (;;) {
        const size_t b = text.find(',', a);
        f.push_back(text.substr(a, b == std::string::npos ? b : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    if (f.size() != 4) throw std::runtime_error("--radio-subchannel expects 4 comma separated fields: '" + text + "'");
    Subchannel sc((subchannel_id_t)id);
    sc.start_address = (subchannel_addr_t)std::stoi(f[0]);
    sc.length = (subchannel_size_t)std::stoi(f[1]);
    if (f[2] == "uep") {
        sc.is_uep = true;
        sc.uep_prot_index = (uep_protection_index_t)std::stoi(f[3]);
    } else {
        const int level = std::stoi(f[2]);
        if (level < 1 || level > 4) throw std::runtime_error("EEP level must be 1..4: '" + text + "'");
        sc.eep_prot_level = (eep_protection_level_t)(level - 1);
        if (f[3] == "A" || f[3] == "a") sc.eep_type = EEP_Type::TYPE_A;
        else if (f[3] == "B" || f[3] == "b") sc.eep_type = EEP_Type::TYPE_B;
        else throw std::runtime_error("EEP type must be A or B: '" + text + "'");
    }
    sc.is_complete = true;
    return sc;
}

static bool parse_args(int argc, char** argv, Args& args) {
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&](void) -> std::string {
            if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
            return argv[++i];
        };
        if (a == "-i" || a == "--input") args.input_file = value();
        else if (a == "--transmission-mode") args.transmission_mode = std::stoi(value());
        else if (a == "--configuration") {
            const std::string c = value();
            if (c == "dab+ofdm") { args.is_ofdm_used = true; args.is_dab_used = true; }
            else if (c == "ofdm") { args.is_ofdm_used = true; args.is_dab_used = false; }
            else if (c == "dab") { args.is_ofdm_used = false; args.is_dab_used = true; }
            else throw std::runtime_error("unknown configuration '" + c + "'");
        }
        else if (a == "--ofdm-input-mode") args.ofdm_input_mode = value();
        else if (a == "--ofdm-block-size") args.ofdm_block_size = (size_t)std::stoull(value());
        else if (a == "--ofdm-total-threads" || a == "--radio-total-threads") (void)value();     // accepted, no effect
        else if (a == "--ofdm-disable-coarse-freq") args.ofdm_disable_coarse_freq = true;
        else if (a == "--ofdm-enable-output") args.ofdm_enable_output = true;
        else if (a == "--ofdm-output") args.ofdm_output = value();
        else if (a == "--ofdm-output-hard-bytes") args.ofdm_output_hard_bytes = true;
        else if (a == "--radio-input-hard-bytes") args.radio_input_hard_bytes = true;
        else if (a == "--radio-fib-output") args.radio_fib_output = value();
        else if (a == "--radio-msc-output") args.radio_msc_output = value();
        else if (a == "--radio-subchannel") { const auto v = value(); args.subchannels.push_back(parse_subchannel(v, args.subchannels.size())); }
        else if (a == "-h" || a == "--help") return false;
        else throw std::runtime_error("unknown argument '" + a + "'");
    }
    if (args.transmission_mode != 1) throw std::runtime_error("only transmission mode I is implemented");
    if (args.ofdm_block_size == 0) throw std::runtime_error("--ofdm-block-size must be positive");
    return true;
}

// BasicRadio's first stage (basic_radio.cpp:41-65, basic_fic_runner.cpp:34-49, basic_dab_plus_channel.cpp:47-51)
class ChannelDecodeStage {
private:
    const DAB_Parameters m_params;
    FIC_Decoder m_fic;
    std::vector<std::unique_ptr<MSC_Decoder>> m_msc;
    FILE* m_fib_out = nullptr;
    std::vector<FILE*> m_msc_out;
public:
    size_t total_frames = 0, total_fibs = 0, total_fib_groups_failed = 0;
    std::vector<size_t> total_msc_bytes;
    ChannelDecodeStage(const Args& args)
    : m_params(get_dab_parameters(args.transmission_mode)),
      m_fic((size_t)m_params.nb_fib_cif_bits, (size_t)m_params.nb_fibs_per_cif) {
        if (!args.radio_fib_output.empty()) {
            m_fib_out = fopen(args.radio_fib_output.c_str(), "wb");
            if (!m_fib_out) throw std::runtime_error("cannot open " + args.radio_fib_output);
        }
        m_fic.OnFIB().Attach([this](tcb::span<const uint8_t> fib) {
            total_fibs++;
            if (m_fib_out) fwrite(fib.data(), 1, fib.size(), m_fib_out);
        });
        for (size_t k = 0; k < args.subchannels.size(); k++) {
            m_msc.push_back(std::make_unique<MSC_Decoder>(args.subchannels[k]));
            const std::string path = args.radio_msc_output + std::to_string(k) + ".bin";
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) throw std::runtime_error("cannot open " + path);
            m_msc_out.push_back(f);
            total_msc_bytes.push_back(0);
        }
    }
    ~ChannelDecodeStage() {
        if (m_fib_out) fclose(m_fib_out);
        for (FILE* f : m_msc_out) fclose(f);
    }
    size_t frame_bits() const { return (size_t)m_params.nb_frame_bits; }
    void Process(tcb::span<const viterbi_bit_t> bits) {
        if (bits.size() != (size_t)m_params.nb_frame_bits) return;                      // basic_radio.cpp:42-47
        total_frames++;
        auto fic_bits = bits.subspan(0, (size_t)m_params.nb_fic_bits);
        auto msc_bits = bits.subspan((size_t)m_params.nb_fic_bits, (size_t)m_params.nb_msc_bits);
        for (int c = 0; c < m_params.nb_cifs; c++) {
            const size_t before = total_fibs;
            m_fic.DecodeFIBGroup(fic_bits.subspan((size_t)c * m_params.nb_fib_cif_bits, (size_t)m_params.nb_fib_cif_bits), (size_t)c);
            if (total_fibs - before != (size_t)m_params.nb_fibs_per_cif) total_fib_groups_failed++;
        }
        for (int c = 0; c < m_params.nb_cifs; c++) {
            auto cif = msc_bits.subspan((size_t)c * m_params.nb_cif_bits, (size_t)m_params.nb_cif_bits);
            for (size_t k = 0; k < m_msc.size(); k++) {
                auto bytes = m_msc[k]->DecodeCIF(cif);
                if (bytes.empty()) continue;                                            // time de-interleaver still filling
                fwrite(bytes.data(), 1, bytes.size(), m_msc_out[k]);
                total_msc_bytes[k] += bytes.size();
            }
        }
    }
};

static int run(const Args& args) {
    FILE* fp_in = stdin;
    if (!args.input_file.empty()) {
        fp_in = fopen(args.input_file.c_str(), "rb");
        if (!fp_in) { fprintf(stderr, "Failed to open input file: '%s'\n", args.input_file.c_str()); return 1; }
    }
    FILE* fp_ofdm_out = nullptr;
    if (args.is_ofdm_used && args.ofdm_enable_output) {
        fp_ofdm_out = stdout;
        if (!args.ofdm_output.empty()) {
            fp_ofdm_out = fopen(args.ofdm_output.c_str(), "wb");
            if (!fp_ofdm_out) { fprintf(stderr, "Failed to open output file: '%s'\n", args.ofdm_output.c_str()); return 1; }
        }
    }
    std::unique_ptr<ChannelDecodeStage> radio;
    if (args.is_dab_used) radio = std::make_unique<ChannelDecodeStage>(args);

    if (args.is_ofdm_used) {
        std::shared_ptr<DeviceIQFileReader> reader;
        try {
            reader = get_iq_file_reader_from_mode_string(fp_in, args.ofdm_input_mode);
        } catch (const std::exception& ex) {
            fprintf(stderr, "Failed to parse OFDM IQ file with format: %s\n%s\n", args.ofdm_input_mode.c_str(), ex.what());
            return 1;
        }
        auto demod = Create_OFDM_Demodulator(args.transmission_mode);
        demod->GetConfig().sync.is_coarse_freq_correction = !args.ofdm_disable_coarse_freq;
        std::vector<uint8_t> hard;
        demod->On_OFDM_Frame().Attach([&](tcb::span<const viterbi_bit_t> bits) {
            if (fp_ofdm_out) {
                if (args.ofdm_output_hard_bytes) {
                    hard.resize(bits.size() / 8);
                    convert_viterbi_bits_to_bytes(bits.first(hard.size() * 8), hard);
                    fwrite(hard.data(), 1, hard.size(), fp_ofdm_out);
                } else {
                    fwrite(bits.data(), 1, bits.size(), fp_ofdm_out);
                }
            }
            if (radio) radio->Process(bits);
        });
        std::vector<std::complex<float>> block(args.ofdm_block_size);               // OFDM_Block::run, app_ofdm_blocks.h:45-57
        for (;;) {
            const size_t length = reader->read(block);
            if (length == 0) break;
            demod->Process(tcb::span<const std::complex<float>>(block.data(), length));
            if (length != block.size()) break;
        }
        demod->Synchronize();
        fprintf(stderr, "ofdm: frames_read=%d frames_desync=%d coarse=%.6g fine=%.6g\n", demod->GetTotalFramesRead(),
                demod->GetTotalFramesDesync(), demod->GetCoarseFrequencyOffset(), demod->GetFineFrequencyOffset());
    } else {
        std::vector<viterbi_bit_t> bits(radio->frame_bits());                         // Basic_Radio_Block::run, app_radio_blocks.h:31-38
        std::vector<uint8_t> packed(bits.size() / 8);
        for (;;) {
            if (args.radio_input_hard_bytes) {
                if (fread(packed.data(), 1, packed.size(), fp_in) != packed.size()) break;
                convert_viterbi_bytes_to_bits(packed, bits);
            } else {
                if (fread(bits.data(), 1, bits.size(), fp_in) != bits.size()) break;
            }
            radio->Process(bits);
        }
    }
    if (radio) {
        fprintf(stderr, "radio: frames=%zu fibs_crc_ok=%zu fib_groups_with_failures=%zu", radio->total_frames, radio->total_fibs,
                radio->total_fib_groups_failed);
        for (size_t k = 0; k < radio->total_msc_bytes.size(); k++) fprintf(stderr, " subchannel%zu_bytes=%zu", k, radio->total_msc_bytes[k]);
        fprintf(stderr, "\n");
    }
    if (fp_ofdm_out && fp_ofdm_out != stdout) fclose(fp_ofdm_out);
    if (fp_in != stdin) fclose(fp_in);
    return 0;
}

int main(int argc, char** argv) {
    Args args;
    try {
        if (!parse_args(argc, argv, args)) { usage(argv[0]); return 0; }
        return run(args);
    } catch (const std::exception& ex) {
        fprintf(stderr, "error: %s\n", ex.what());
        return 1;
    }
}
