// dab/dabgpu_frame_batcher.cpp -- see dabgpu_frame_batcher.h
#include "./dabgpu_frame_batcher.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "./dabgpu_shared_context.h"

namespace dabgpu_frame_batcher {
namespace {

constexpr int KEEP = 8;                                    // frames whose soft bits and results stay available (= the session's)
constexpr int MAX_PRODUCERS = 64;                          // demodulators with a session of their own; further ones are not batched
constexpr size_t FRAME_BITS = DABGPU_NB_FRAME_BITS, FIC_BITS = 9216, CIF_BITS = 55296, GROUP_BITS = 2304;

// one demodulator's frames: its receiver's frame session (owned by the demodulator) + the host copy of its last KEEP frames (what the
// decoders' buffers are matched against)
// Locking (one process-wide mutex used to be held over every memcmp, every 230 KB frame copy and every call into a session, so the
// decoders of ALL receivers of the process serialised there): State::mu guards the subscription and which producer slot belongs to whom;
// every Producer has a mutex of its own for its frames and its session pointer.  Order: State::mu before Producer::mu, never the reverse;
// the fetch / match paths take a Producer::mu only.
struct Producer {
    std::mutex mu;
    std::atomic<const void*> id{nullptr};                  // assigned under State::mu
    dabgpu_frame_session* session = nullptr;
    struct Frame { uint64_t gen = ~0ull; std::vector<int8_t> bits; };
    Frame frames[KEEP];
    uint64_t next_gen = 0;
};

struct State {
    std::mutex mu;
    int fic_refs = 0;
    struct Sub { dabgpu_subchannel sc; int refs; };
    std::vector<Sub> subs;
    uint64_t subs_version = 0;
    Producer producers[MAX_PRODUCERS];                     // index = cif_id::src; slots are never moved
    std::atomic<int> n_producers{0};                       // slots ever handed out
};
State& S() { static State s; return s; }
std::atomic<unsigned long long> n_fib_batched{0}, n_fib_own{0}, n_cif_batched{0}, n_cif_own{0};

bool same(const dabgpu_subchannel& a, const dabgpu_subchannel& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

void forget(Producer& p) {
    p.session = nullptr;                                   // (the session is the demodulator's)
    for (auto& f : p.frames) f.gen = ~0ull;
    p.next_gen = 0;
}

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
    s.subs_version++;
}

void remove_subchannel(const dabgpu_subchannel& sc) {
    if (!enabled()) return;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (size_t k = 0; k < s.subs.size(); k++)
        if (same(s.subs[k].sc, sc)) {
            if (--s.subs[k].refs == 0) { s.subs.erase(s.subs.begin() + (std::ptrdiff_t)k); s.subs_version++; }
            return;
        }
}

void remove_producer(const void* producer) {
    if (!enabled()) return;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (int k = 0; k < s.n_producers.load(); k++) {
        Producer& p = s.producers[k];
        if (p.id.load() != producer) continue;
        std::lock_guard<std::mutex> gp(p.mu);                            // (a fetch that is inside the session finishes first: the session dies after this returns)
        forget(p);
        p.id.store(nullptr);                                             // the slot stays (cif_id::src of others must not move)
    }
}

uint64_t subscription(std::vector<dabgpu_subchannel>& subs, bool& fic) {
    subs.clear();
    fic = false;
    if (!enabled()) return 0;
    State& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (const auto& e : s.subs) subs.push_back(e.sc);
    fic = s.fic_refs > 0;
    return (s.subs_version << 1) | (fic ? 1u : 0u);
}

void on_frame_decoded(const void* producer, dabgpu_frame_session* session, uint64_t gen, const int8_t* frame_bits) {
    if (!enabled() || !session) return;
    State& s = S();
    Producer* p = nullptr;
    {
        std::lock_guard<std::mutex> g(s.mu);
        if (s.fic_refs == 0 && s.subs.empty()) return;                   // nobody is listening
        const int n = s.n_producers.load();
        for (int k = 0; k < n && !p; k++) if (s.producers[k].id.load() == producer) p = &s.producers[k];
        if (!p) {
            for (int k = 0; k < n && !p; k++) if (s.producers[k].id.load() == nullptr) p = &s.producers[k];      // a slot a destroyed demodulator left
            if (!p) {
                if (n >= MAX_PRODUCERS) return;                          // this demodulator's decoders decode call by call
                p = &s.producers[n];
                s.n_producers.store(n + 1);
            }
            p->id.store(producer);
        }
    }
    // (this demodulator's own delivery thread is the only writer of its slot, and remove_producer runs in its destructor, after that thread)
    std::lock_guard<std::mutex> gp(p->mu);
    p->session = session;
    Producer::Frame& f = p->frames[gen % KEEP];
    f.gen = ~0ull;
    f.bits.assign(frame_bits, frame_bits + FRAME_BITS);
    f.gen = gen;
    p->next_gen = gen + 1;
}

bool fetch_fib_group(const int8_t* group_bits, int group, uint8_t* bytes96, uint32_t* crc_mask, uint64_t* path_error) {
    if (!enabled() || group < 0 || group > 3) return false;
    State& s = S();
    for (int k = 0, n = s.n_producers.load(); k < n; k++) {
        Producer& p = s.producers[k];
        std::lock_guard<std::mutex> g(p.mu);                              // (held over the fetch -- a copy out of host memory, the frame was delivered complete: remove_producer forgets sessions)
        if (!p.session) continue;
        for (uint64_t back = 0; back < KEEP && back < p.next_gen; back++) {          // newest first
            const Producer::Frame& f = p.frames[(p.next_gen - 1 - back) % KEEP];
            if (f.gen != p.next_gen - 1 - back) continue;
            if (std::memcmp(f.bits.data() + (size_t)group * GROUP_BITS, group_bits, GROUP_BITS) == 0) {
                const bool ok = dabgpu_frame_session_fetch_fib_group(p.session, f.gen, group, bytes96, crc_mask, path_error) == DABGPU_OK;
                if (ok) n_fib_batched++;
                return ok;
            }
        }
    }
    return false;
}

cif_id match_cif(const int8_t* slice_bits, size_t start_bit, size_t n_bits, cif_id after) {
    cif_id none;
    if (!enabled() || start_bit + n_bits > CIF_BITS) return none;
    State& s = S();
    const int n_prod = s.n_producers.load();
    auto holds = [&](cif_id id) {
        if (id.src < 0 || id.src >= n_prod) return false;
        Producer& p = s.producers[id.src];
        std::lock_guard<std::mutex> g(p.mu);
        if (!p.session) return false;
        if (id.gen >= p.next_gen || p.next_gen - id.gen > KEEP) return false;
        const Producer::Frame& f = p.frames[id.gen % KEEP];
        return f.gen == id.gen && std::memcmp(f.bits.data() + FIC_BITS + (size_t)id.cif * CIF_BITS + start_bit, slice_bits, n_bits) == 0;
    };
    // a decoder follows one demodulator: the successor of the CIF it consumed last is tried first (one memcmp in the steady state) ...
    if (after.valid()) { const cif_id nx = successor(after); if (holds(nx)) return nx; }
    // ... then the frames of the demodulator it followed, then everybody else's
    for (int pass = 0; pass < 2; pass++)
        for (int src = 0; src < n_prod; src++) {
            if ((pass == 0) != (after.valid() && src == after.src)) continue;
            uint64_t next_gen;
            { Producer& p = s.producers[src]; std::lock_guard<std::mutex> g(p.mu); if (!p.session) continue; next_gen = p.next_gen; }
            for (uint64_t back = 0; back < KEEP && back < next_gen; back++)
                for (int c = 0; c < 4; c++) { cif_id id; id.src = src; id.gen = next_gen - 1 - back; id.cif = c; if (holds(id)) return id; }
        }
    return none;
}

bool fetch_cif(cif_id id, const dabgpu_subchannel& sc, uint8_t* bytes, size_t capacity, size_t* n_bytes, uint64_t* path_error) {
    State& s = S();
    if (id.src < 0 || id.src >= s.n_producers.load()) return false;
    Producer& p = s.producers[id.src];
    std::lock_guard<std::mutex> g(p.mu);
    if (!p.session) return false;
    const bool ok = dabgpu_frame_session_fetch_cif(p.session, id.gen, &sc, id.cif, bytes, capacity, n_bytes, path_error) == DABGPU_OK;
    if (ok) n_cif_batched++;
    return ok;
}

Counters counters() { return Counters{n_fib_batched.load(), n_fib_own.load(), n_cif_batched.load(), n_cif_own.load()}; }
void count_call_by_call(bool fib_group) { if (fib_group) n_fib_own++; else n_cif_own++; }

}  // namespace dabgpu_frame_batcher
