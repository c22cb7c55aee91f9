// fake_dabgpu_oracle.cpp -- TEST INFRASTRUCTURE ONLY, never linked into the product: the entry points of include/dabgpu.h that the
// C++ mirror classes call (dab-radio_amd/host/**), implemented on the CPU oracle (oracle/*.c), so that the classes' own host logic --
// the framing state machine, the process-wide frame batcher with its per-demodulator sessions, the shared context, the decoders'
// bookkeeping -- can run under ThreadSanitizer / AddressSanitizer on a machine without a GPU (tests/test_host_sanitizers.py).
// Results are the oracle's, i.e. the ones the device produces (that equality is what the -m gpu tests establish); here only the
// threading and memory behaviour of the host code is under test.
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "dabgpu.h"
extern "C" {
#include "dab_oracle.h"
}

struct dabgpu_ctx { std::vector<int> mapper; };

extern "C" {

int dabgpu_device_count(void) { return 1; }
int dabgpu_create(dabgpu_ctx** out, int, const float*, const int* h_mapper) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    dabgpu_ctx* c = new dabgpu_ctx();
    c->mapper.resize(DAB_NB_DATA_CARRIERS);
    if (h_mapper) std::memcpy(c->mapper.data(), h_mapper, sizeof(int) * DAB_NB_DATA_CARRIERS); else dab_get_mapper(c->mapper.data());
    *out = c;
    return DABGPU_OK;
}
void dabgpu_destroy(dabgpu_ctx* c) { delete c; }
int dabgpu_host_pin(void*, size_t) { return DABGPU_OK; }
int dabgpu_host_unpin(void*) { return DABGPU_OK; }

int dabgpu_ofdm_sync_host_sync_mode(dabgpu_ctx* c, int mode, const float* h_prs_sym, const dabgpu_sync_cfg* cfg, dabgpu_sync_state* st, float* imp, float* frq) {
    if (!c || !h_prs_sym || !cfg || !st) return DABGPU_ERR_INVALID_ARG;
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return DABGPU_ERR_INVALID_ARG;
    std::vector<dab_cf32> prs((size_t)g.nb_fft), conj_ref((size_t)g.nb_fft), time_ref((size_t)g.nb_fft);
    dab_get_prs_fft_mode(mode, prs.data());
    dab_sync_refs_mode(mode, prs.data(), conj_ref.data(), time_ref.data());
    dab_sync_cfg oc;
    oc.fine_freq_update_beta = cfg->fine_freq_update_beta; oc.is_coarse_freq_correction = cfg->is_coarse_freq_correction;
    oc.max_coarse_freq_correction_norm = cfg->max_coarse_freq_correction_norm; oc.coarse_freq_slow_beta = cfg->coarse_freq_slow_beta;
    oc.impulse_peak_threshold_db = cfg->impulse_peak_threshold_db; oc.impulse_peak_distance_probability = cfg->impulse_peak_distance_probability;
    dab_sync_state os = {st->freq_coarse, st->freq_fine, st->is_found_coarse, 0, 0, 0};
    const dab_cf32* sym = reinterpret_cast<const dab_cf32*>(h_prs_sym);
    dab_coarse_freq_sync_mode(mode, sym, time_ref.data(), &oc, &os, frq);
    int off = 0;
    const int ok = dab_fine_time_sync_mode(mode, sym, conj_ref.data(), &oc, os.freq_coarse + os.freq_fine, &off, imp);
    st->freq_coarse = os.freq_coarse; st->freq_fine = os.freq_fine; st->is_found_coarse = os.is_found_coarse;
    st->sync_valid = ok; if (ok) st->fine_time_offset = off;
    return DABGPU_OK;
}

int dabgpu_ofdm_demod_stream_frame_sync(dabgpu_ctx* c, const float* h_iq, float coarse, float* fine, float beta, int8_t* h_bits, float* h_total, float* h_fft, float*) {
    if (!c || !h_iq || !fine || !h_bits) return DABGPU_ERR_INVALID_ARG;
    const float total = dab_demod_frame(reinterpret_cast<const dab_cf32*>(h_iq), coarse + *fine, c->mapper.data(), h_bits, nullptr, nullptr, reinterpret_cast<dab_cf32*>(h_fft));
    *fine = dab_update_fine_freq_mode(1, *fine, total, beta);
    if (h_total) *h_total = total;
    return DABGPU_OK;
}
int dabgpu_ofdm_demod_stream_frame_sync_mode(dabgpu_ctx* c, int mode, const float* h_iq, float coarse, float* fine, float beta, int8_t* h_bits, float* h_total, float* h_fft) {
    if (!c || !h_iq || !fine || !h_bits) return DABGPU_ERR_INVALID_ARG;
    dab_ofdm_geometry g;
    if (dab_ofdm_geometry_get(mode, &g)) return DABGPU_ERR_INVALID_ARG;
    std::vector<int> map((size_t)g.nb_carriers);
    dab_mapper_n(g.nb_fft, g.nb_carriers, map.data());
    const float total = dab_demod_frame_mode(mode, reinterpret_cast<const dab_cf32*>(h_iq), coarse + *fine, map.data(), h_bits, nullptr, nullptr, reinterpret_cast<dab_cf32*>(h_fft));
    *fine = dab_update_fine_freq_mode(mode, *fine, total, beta);
    if (h_total) *h_total = total;
    return DABGPU_OK;
}

int dabgpu_fic_decode_group_host_sync(dabgpu_ctx* c, const int8_t* h_bits, uint8_t* h_bytes, uint32_t* mask, uint64_t* err, int tie) {
    if (!c || !h_bits || !h_bytes) return DABGPU_ERR_INVALID_ARG;
    uint32_t m = 0;
    const uint64_t e = dab_fic_decode_group(h_bits, tie, h_bytes, &m);
    if (mask) *mask = m;
    if (err) *err = e;
    return DABGPU_OK;
}

int dabgpu_viterbi_decode_depunctured_host_sync(dabgpu_ctx* c, const int8_t* h_mother, size_t n_steps, uint32_t start, uint32_t end, uint8_t* h_out, size_t n_out,
                                                uint64_t* err, int tie) {
    if (!c || !h_mother || n_steps < 1 || (n_out && n_out * 8 + 6 > n_steps)) return DABGPU_ERR_INVALID_ARG;
    dab_viterbi* v = dab_viterbi_create(n_steps + 8, tie);
    dab_viterbi_reset(v, start);
    const uint8_t all[1] = {4};
    dab_viterbi_update(v, h_mother, 4 * n_steps, all, 1, 4 * n_steps);
    const uint64_t e = dab_viterbi_chainback(v, h_out, n_out, end);
    if (err) *err = e;
    dab_viterbi_destroy(v);
    return DABGPU_OK;
}

// ---- per-sub-channel stream (MSC_Decoder, CIF_Deinterleaver) ----
struct dabgpu_msc_stream { dab_subchannel sc; dab_deinterleaver* d; int nbits; std::vector<int8_t> logical; bool ready; };
static dab_subchannel to_oracle(const dabgpu_subchannel& s) { return dab_subchannel{s.start_address, s.length, s.is_uep, s.uep_prot_index, s.eep_prot_level, s.eep_type}; }
int dabgpu_msc_stream_create(dabgpu_ctx* c, const dabgpu_subchannel* sc, dabgpu_msc_stream** out) {
    if (!c || !sc || !out) return DABGPU_ERR_INVALID_ARG;
    int pi[4], lx[4], nb;
    dab_subchannel o = to_oracle(*sc);
    if (sc->length <= 0 || sc->length > 864 || dab_subchannel_plan(&o, pi, lx, &nb) < 0) return DABGPU_ERR_INVALID_ARG;
    dabgpu_msc_stream* s = new dabgpu_msc_stream();
    s->sc = o; s->nbits = sc->length * 64; s->d = dab_deinterleaver_create(sc->length * 8); s->logical.resize((size_t)s->nbits); s->ready = false;
    *out = s;
    return DABGPU_OK;
}
void dabgpu_msc_stream_destroy(dabgpu_msc_stream* s) { if (!s) return; dab_deinterleaver_destroy(s->d); delete s; }
int dabgpu_msc_stream_push_cif(dabgpu_msc_stream* s, const int8_t* h_bits) {
    if (!s || !h_bits) return DABGPU_ERR_INVALID_ARG;
    dab_deinterleaver_consume(s->d, h_bits);
    s->ready = dab_deinterleaver_deinterleave(s->d, s->logical.data()) != 0;
    return DABGPU_OK;
}
int dabgpu_msc_stream_deinterleave_sync(dabgpu_msc_stream* s, int8_t* h_out) {
    if (!s || !h_out) return DABGPU_ERR_INVALID_ARG;
    if (!s->ready) return DABGPU_ERR_NOT_READY;
    std::memcpy(h_out, s->logical.data(), (size_t)s->nbits);
    return DABGPU_OK;
}
int dabgpu_msc_stream_decode_sync(dabgpu_msc_stream* s, uint8_t* h_out, size_t* n_out, uint64_t* err, int tie) {
    if (!s || !h_out || !n_out) return DABGPU_ERR_INVALID_ARG;
    *n_out = 0;
    if (!s->ready) return DABGPU_ERR_NOT_READY;
    int nb = 0;
    const uint64_t e = dab_msc_decode_logical(&s->sc, s->logical.data(), tie, h_out, &nb);
    *n_out = (size_t)nb;
    if (err) *err = e;
    return DABGPU_OK;
}

// ---- frame session: one decode per pushed frame, results of the last 8 frames ----
struct dabgpu_frame_session {
    static constexpr int R = 8;
    std::mutex mu;
    struct Sub { dabgpu_subchannel sc; dab_deinterleaver* d; int cifs; };
    std::vector<Sub> subs;
    struct Slot {
        uint64_t gen = ~0ull; bool fic = false;
        uint8_t fib[4][96]; uint32_t mask[4]; uint64_t ferr[4];
        struct Out { dabgpu_subchannel sc; std::vector<uint8_t> bytes[4]; uint64_t err[4]; bool ok[4]; };
        std::vector<Out> outs;
    } slots[R];
    uint64_t next = 0;
};
int dabgpu_frame_session_create(dabgpu_frame_session** out, int) { if (!out) return DABGPU_ERR_INVALID_ARG; *out = new dabgpu_frame_session(); return DABGPU_OK; }
void dabgpu_frame_session_destroy(dabgpu_frame_session* s) {
    if (!s) return;
    for (auto& e : s->subs) dab_deinterleaver_destroy(e.d);
    delete s;
}
int dabgpu_frame_session_set_subchannels(dabgpu_frame_session* s, const dabgpu_subchannel* subs, int n) {
    if (!s || n < 0 || n > 64 || (n && !subs)) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    std::vector<dabgpu_frame_session::Sub> keep;
    for (int k = 0; k < n; k++) {
        bool found = false;
        for (auto& e : s->subs) if (e.d && !std::memcmp(&e.sc, &subs[k], sizeof(subs[k]))) { keep.push_back(e); e.d = nullptr; found = true; break; }
        if (!found) keep.push_back({subs[k], dab_deinterleaver_create(subs[k].length * 8), 0});
    }
    for (auto& e : s->subs) if (e.d) dab_deinterleaver_destroy(e.d);
    s->subs = keep;
    return DABGPU_OK;
}
int dabgpu_frame_session_push_frame(dabgpu_frame_session* s, const int8_t* bits, int decode_fic, int tie, uint64_t* generation) {
    if (!s || !bits) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    const uint64_t gen = s->next++;
    auto& sl = s->slots[gen % dabgpu_frame_session::R];
    sl.gen = gen; sl.fic = decode_fic != 0; sl.outs.clear();
    if (decode_fic) for (int k = 0; k < 4; k++) sl.ferr[k] = dab_fic_decode_group(bits + k * 2304, tie, sl.fib[k], &sl.mask[k]);
    for (auto& e : s->subs) {
        dabgpu_frame_session::Slot::Out o;
        o.sc = e.sc;
        dab_subchannel osc = to_oracle(e.sc);
        std::vector<int8_t> logical((size_t)e.sc.length * 64);
        for (int c = 0; c < 4; c++) {
            dab_deinterleaver_consume(e.d, bits + 9216 + (size_t)c * 55296 + (size_t)e.sc.start_address * 64);
            o.ok[c] = dab_deinterleaver_deinterleave(e.d, logical.data()) != 0;
            o.err[c] = 0;
            if (o.ok[c]) { o.bytes[c].resize((size_t)e.sc.length * 8); int nb = 0; o.err[c] = dab_msc_decode_logical(&osc, logical.data(), tie, o.bytes[c].data(), &nb); o.bytes[c].resize((size_t)nb); }
        }
        sl.outs.push_back(std::move(o));
    }
    if (generation) *generation = gen;
    return DABGPU_OK;
}
int dabgpu_frame_session_fetch_fib_group(dabgpu_frame_session* s, uint64_t gen, int group, uint8_t* bytes, uint32_t* mask, uint64_t* err) {
    if (!s || group < 0 || group > 3 || !bytes) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    auto& sl = s->slots[gen % dabgpu_frame_session::R];
    if (sl.gen != gen || !sl.fic) return DABGPU_ERR_NOT_READY;
    std::memcpy(bytes, sl.fib[group], 96);
    if (mask) *mask = sl.mask[group];
    if (err) *err = sl.ferr[group];
    return DABGPU_OK;
}
int dabgpu_frame_session_fetch_cif(dabgpu_frame_session* s, uint64_t gen, const dabgpu_subchannel* sc, int cif, uint8_t* bytes, size_t cap, size_t* n, uint64_t* err) {
    if (!s || !sc || cif < 0 || cif > 3 || !bytes || !n) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(s->mu);
    auto& sl = s->slots[gen % dabgpu_frame_session::R];
    if (sl.gen != gen) return DABGPU_ERR_NOT_READY;
    for (auto& o : sl.outs)
        if (!std::memcmp(&o.sc, sc, sizeof(*sc))) {
            if (!o.ok[cif] || o.bytes[cif].size() > cap) return DABGPU_ERR_NOT_READY;
            std::memcpy(bytes, o.bytes[cif].data(), o.bytes[cif].size());
            *n = o.bytes[cif].size();
            if (err) *err = o.err[cif];
            return DABGPU_OK;
        }
    return DABGPU_ERR_NOT_READY;
}


// ---- receiver pipeline: computed at submission (the host code's queues, threads and waits are what runs under the sanitizers) ----
struct dabgpu_receiver {
    int mode = 1;
    dab_ofdm_geometry g;
    dabgpu_ctx* ctx = nullptr;
    dabgpu_frame_session* ses = nullptr;
    std::vector<float> stage[3];
    int cur = 0;
    dabgpu_sync_state state = {0, 0, 0, 0, 0, 0};
    dabgpu_sync_state record = {0, 0, 0, 0, 0, 0};
    std::vector<float> imp, frq;
    bool sync_pending = false, sync_coarse = false;
    std::atomic<int> decode_fic{0};
    std::mutex mu;                                       // result slots: written by the reader thread, read by the delivery thread
    struct Slot { uint64_t gen = ~0ull; std::vector<int8_t> bits; float fine = 0, total = 0; std::vector<float> fft, dq; } slots[8];
    uint64_t next = 0;
};
int dabgpu_receiver_create(dabgpu_receiver** out, int, int mode, const float*, const int* h_mapper) {
    if (!out) return DABGPU_ERR_INVALID_ARG;
    dabgpu_receiver* rx = new dabgpu_receiver();
    if (dab_ofdm_geometry_get(mode, &rx->g)) { delete rx; return DABGPU_ERR_INVALID_ARG; }
    rx->mode = mode;
    dabgpu_create(&rx->ctx, 0, nullptr, mode == 1 ? h_mapper : nullptr);
    dabgpu_frame_session_create(&rx->ses, 0);
    const size_t cap = (size_t)rx->g.nb_null_period + (size_t)(rx->g.nb_fft - rx->g.nb_cp) + (size_t)rx->g.nb_frame_symbols * rx->g.nb_symbol_period + rx->g.nb_null_period;
    for (auto& st : rx->stage) st.assign(2 * cap, 0.0f);
    rx->imp.assign((size_t)rx->g.nb_fft, 0.0f); rx->frq.assign((size_t)rx->g.nb_fft, 0.0f);
    *out = rx;
    return DABGPU_OK;
}
int dabgpu_receiver_create_banked(dabgpu_receiver** out, int device) { return dabgpu_receiver_create(out, device, 1, nullptr, nullptr); }   // (no streams to share here)
void dabgpu_receiver_destroy(dabgpu_receiver* rx) { if (!rx) return; dabgpu_frame_session_destroy(rx->ses); dabgpu_destroy(rx->ctx); delete rx; }
dabgpu_frame_session* dabgpu_receiver_session(dabgpu_receiver* rx) { return rx ? rx->ses : nullptr; }
int dabgpu_receiver_set_subchannels(dabgpu_receiver* rx, const dabgpu_subchannel* subs, int n, int decode_fic) {
    if (!rx) return DABGPU_ERR_INVALID_ARG;
    const int st = dabgpu_frame_session_set_subchannels(rx->ses, subs, n);
    if (st) return st;
    rx->decode_fic = decode_fic;
    return DABGPU_OK;
}
int dabgpu_receiver_stage(dabgpu_receiver* rx, float** h, size_t* cap) {
    if (!rx || !h) return DABGPU_ERR_INVALID_ARG;
    *h = rx->stage[rx->cur].data();
    if (cap) *cap = rx->stage[rx->cur].size() / 2;
    return DABGPU_OK;
}
int dabgpu_receiver_reset(dabgpu_receiver* rx) { if (!rx) return DABGPU_ERR_INVALID_ARG; rx->state = {0, 0, 0, 0, 0, 0}; return DABGPU_OK; }
int dabgpu_receiver_submit_sync(dabgpu_receiver* rx, const dabgpu_sync_cfg* cfg, size_t prs_sample) {
    if (!rx || !cfg || rx->sync_pending) return DABGPU_ERR_INVALID_ARG;
    rx->sync_coarse = cfg->is_coarse_freq_correction != 0;
    const int st = dabgpu_ofdm_sync_host_sync_mode(rx->ctx, rx->mode, rx->stage[rx->cur].data() + 2 * prs_sample, cfg, &rx->state, rx->imp.data(),
                                                   rx->sync_coarse ? rx->frq.data() : nullptr);
    rx->record = rx->state;
    rx->sync_pending = true;
    return st;
}
int dabgpu_receiver_wait_sync(dabgpu_receiver* rx, dabgpu_sync_state* out, float* imp, float* frq) {
    if (!rx || !out) return DABGPU_ERR_INVALID_ARG;
    if (!rx->sync_pending) return DABGPU_ERR_NOT_READY;
    rx->sync_pending = false;
    *out = rx->record;
    if (imp) std::memcpy(imp, rx->imp.data(), rx->imp.size() * sizeof(float));
    if (frq && rx->sync_coarse) std::memcpy(frq, rx->frq.data(), rx->frq.size() * sizeof(float));
    return DABGPU_OK;
}
int dabgpu_receiver_submit_demod(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, uint64_t* generation) {
    if (!rx || rx->sync_pending) return DABGPU_ERR_INVALID_ARG;
    const size_t n_bits = (size_t)(rx->g.nb_frame_symbols - 1) * 2 * rx->g.nb_carriers;
    std::vector<int8_t> bits(n_bits);
    std::vector<float> fft(want_views ? (size_t)(rx->g.nb_frame_symbols + 1) * rx->g.nb_fft * 2 : 0);
    float fine = rx->state.freq_fine, total = 0;
    const float* iq = rx->stage[rx->cur].data() + 2 * frame_sample;
    const int st = rx->mode == 1 ? dabgpu_ofdm_demod_stream_frame_sync(rx->ctx, iq, rx->state.freq_coarse, &fine, beta, bits.data(), &total, want_views ? fft.data() : nullptr, nullptr)
                                 : dabgpu_ofdm_demod_stream_frame_sync_mode(rx->ctx, rx->mode, iq, rx->state.freq_coarse, &fine, beta, bits.data(), &total, want_views ? fft.data() : nullptr);
    if (st) return st;
    rx->state.freq_fine = fine;
    uint64_t gen;
    {
        std::lock_guard<std::mutex> g(rx->mu);
        gen = rx->next;
        auto& sl = rx->slots[gen % 8];
        sl.gen = gen; sl.bits = std::move(bits); sl.fine = fine; sl.total = total; sl.fft = std::move(fft);
        sl.dq.assign(want_views && rx->mode == 1 ? (size_t)(rx->g.nb_frame_symbols - 1) * rx->g.nb_carriers * 2 : 0, 0.0f);
        rx->next = gen + 1;
    }
    rx->cur = (rx->cur + 1) % 3;
    if (generation) *generation = gen;
    return DABGPU_OK;
}
// the decode of a demodulated frame: the frame session's push (its generations count the same frames)
int dabgpu_receiver_submit_decode(dabgpu_receiver* rx, uint64_t gen, int tie) {
    if (!rx) return DABGPU_ERR_INVALID_ARG;
    if (rx->mode != 1) return DABGPU_OK;
    std::vector<int8_t> bits;
    { std::lock_guard<std::mutex> g(rx->mu); if (rx->slots[gen % 8].gen != gen) return DABGPU_ERR_NOT_READY; bits = rx->slots[gen % 8].bits; }
    uint64_t sgen = 0;
    if (dabgpu_frame_session_push_frame(rx->ses, bits.data(), rx->decode_fic, tie, &sgen) || sgen != gen) return DABGPU_ERR_HIP;
    return DABGPU_OK;
}
int dabgpu_receiver_submit_frame(dabgpu_receiver* rx, size_t frame_sample, float beta, int want_views, int tie, uint64_t* generation) {
    uint64_t gen = 0;
    int st = dabgpu_receiver_submit_demod(rx, frame_sample, beta, want_views, &gen);
    if (!st) st = dabgpu_receiver_submit_decode(rx, gen, tie);
    if (!st && generation) *generation = gen;
    return st;
}
int dabgpu_receiver_wait_frame(dabgpu_receiver* rx, uint64_t gen, dabgpu_receiver_frame* out) {
    if (!rx || !out) return DABGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(rx->mu);
    auto& sl = rx->slots[gen % 8];
    if (sl.gen != gen) return DABGPU_ERR_NOT_READY;
    out->generation = gen; out->bits = sl.bits.data(); out->n_bits = sl.bits.size(); out->freq_fine = sl.fine; out->total_phase = sl.total;
    out->fft = sl.fft.empty() ? nullptr : sl.fft.data(); out->dqpsk = sl.dq.empty() ? nullptr : sl.dq.data();
    return DABGPU_OK;
}

}  // extern "C"
