// dab/dabgpu_frame_batcher.cpp -- see dabgpu_frame_batcher.h
#include "./dabgpu_frame_batcher.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "./dabgpu_shared_context.h"

namespace dabgpu_frame_batcher {
namespace {

constexpr int KEEP = 8;                                    // frames whose soft bits and results stay available (= the session's)
constexpr size_t FRAME_BITS = DABGPU_NB_FRAME_BITS, FIC_BITS = 9216, CIF_BITS = 55296, GROUP_BITS = 2304;

struct State {
    std::mutex mu;
    dabgpu_frame_session* session = nullptr;
    bool failed = false;                                   // the session could not be created / a push failed: stay out of the way
    int fic_refs = 0;
    struct Sub { dabgpu_subchannel sc; int refs; };
    std::vector<Sub> subs;
    bool subs_dirty = false;
    struct Frame { uint64_t gen = ~0ull; std::vector<int8_t> bits; };
    Frame frames[KEEP];
    uint64_t next_gen = 0;
};
State& S() { static State s; return s; }

bool same(const dabgpu_subchannel& a, const dabgpu_subchannel& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

}  // namespace

bool enabled() {
    static const bool on = [] { const char* e = std::getenv("DABGPU_MIRROR_BATCH"); return !(e && std::atoi(e) == 0); }();
    return on;
}

void add_fic() { if (!enabled()) return; std::lock_guard<std::mutex> g(S().mu); S().fic_refs++; }
void remove_fic() { if (!enabled()) return; std::lock_guard<std::mutex> g(S().mu); if (S().fic_refs > 0) S().fic_refs--; }

void add_subchannel(const dabgpu_subchannel& sc) {
    if (!enabled()) return;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (auto& e : s.subs) if (same(e.sc, sc)) { e.refs++; return; }
    if (s.subs.size() >= 64) return;                       // more than the session takes: the extra decoders decode on their own
    s.subs.push_back({sc, 1});
    s.subs_dirty = true;
}

void remove_subchannel(const dabgpu_subchannel& sc) {
    if (!enabled()) return;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (size_t k = 0; k < s.subs.size(); k++)
        if (same(s.subs[k].sc, sc)) {
            if (--s.subs[k].refs == 0) { s.subs.erase(s.subs.begin() + (std::ptrdiff_t)k); s.subs_dirty = true; }
            return;
        }
}

void on_frame(const int8_t* frame_bits) {
    if (!enabled()) return;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    if (s.failed || (s.fic_refs == 0 && s.subs.empty())) return;         // nobody is listening
    if (!s.session) {
        const char* dev = std::getenv("DABGPU_DEVICE");
        if (dabgpu_frame_session_create(&s.session, dev ? std::atoi(dev) : 0) != DABGPU_OK) { s.failed = true; return; }
        s.subs_dirty = true;
    }
    if (s.subs_dirty) {
        std::vector<dabgpu_subchannel> list;
        for (const auto& e : s.subs) list.push_back(e.sc);
        if (dabgpu_frame_session_set_subchannels(s.session, list.data(), (int)list.size()) != DABGPU_OK) { s.failed = true; return; }
        s.subs_dirty = false;
    }
    uint64_t gen = 0;
    if (dabgpu_frame_session_push_frame(s.session, frame_bits, s.fic_refs > 0, dabgpu_tie_rule_from_env(), &gen) != DABGPU_OK) { s.failed = true; return; }
    State::Frame& f = s.frames[gen % KEEP];
    f.gen = ~0ull;
    f.bits.assign(frame_bits, frame_bits + FRAME_BITS);
    f.gen = gen;
    s.next_gen = gen + 1;
}

bool fetch_fib_group(const int8_t* group_bits, int group, uint8_t* bytes96, uint32_t* crc_mask, uint64_t* path_error) {
    if (!enabled() || group < 0 || group > 3) return false;
    State& s = S();
    uint64_t gen = ~0ull;
    {
        std::lock_guard<std::mutex> g(s.mu);
        if (!s.session || s.failed) return false;
        for (uint64_t back = 0; back < KEEP && back < s.next_gen; back++) {          // newest first
            const State::Frame& f = s.frames[(s.next_gen - 1 - back) % KEEP];
            if (f.gen != s.next_gen - 1 - back) continue;
            if (std::memcmp(f.bits.data() + (size_t)group * GROUP_BITS, group_bits, GROUP_BITS) == 0) { gen = f.gen; break; }
        }
    }
    if (gen == ~0ull) return false;
    return dabgpu_frame_session_fetch_fib_group(s.session, gen, group, bytes96, crc_mask, path_error) == DABGPU_OK;
}

cif_id match_cif(const int8_t* slice_bits, size_t start_bit, size_t n_bits, cif_id after) {
    cif_id none;
    if (!enabled() || start_bit + n_bits > CIF_BITS) return none;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.session || s.failed || s.next_gen == 0) return none;
    auto holds = [&](cif_id id) {
        if (id.gen >= s.next_gen || s.next_gen - id.gen > KEEP) return false;
        const State::Frame& f = s.frames[id.gen % KEEP];
        return f.gen == id.gen && std::memcmp(f.bits.data() + FIC_BITS + (size_t)id.cif * CIF_BITS + start_bit, slice_bits, n_bits) == 0;
    };
    if (after.valid()) { const cif_id nx = successor(after); if (holds(nx)) return nx; }
    for (uint64_t back = 0; back < KEEP && back < s.next_gen; back++)
        for (int c = 0; c < 4; c++) { cif_id id; id.gen = s.next_gen - 1 - back; id.cif = c; if (holds(id)) return id; }
    return none;
}

bool fetch_cif(cif_id id, const dabgpu_subchannel& sc, uint8_t* bytes, size_t capacity, size_t* n_bytes, uint64_t* path_error) {
    State& s = S();
    {
        std::lock_guard<std::mutex> g(s.mu);
        if (!s.session || s.failed) return false;
    }
    return dabgpu_frame_session_fetch_cif(s.session, id.gen, &sc, id.cif, bytes, capacity, n_bytes, path_error) == DABGPU_OK;
}

}  // namespace dabgpu_frame_batcher
