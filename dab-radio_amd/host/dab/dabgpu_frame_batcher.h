// dab/dabgpu_frame_batcher.h -- one batched device decode per transmission frame behind FIC_Decoder / MSC_Decoder.
//
// The reference fans a frame out to a FIC runner and one MSC runner per sub-channel (src/basic_radio/basic_radio.cpp:41-65); behind
// the mirror classes every DecodeFIBGroup / DecodeCIF used to be its own synchronous launch + two copies: 4 + 72 round trips per frame
// for 18 sub-channels.  The batcher is a process-wide registry between the two sides: every FIC_Decoder / MSC_Decoder registers what
// it decodes, every OFDM_Demod (mode I) asks it what to decode (`subscription`) and has its receiver pipeline decode exactly that with
// the frame, ON THE DEVICE, chained behind the demodulation (dabgpu_receiver_*, include/dabgpu.h) -- the FIC and all registered
// sub-channels in one batched decode; when the frame is delivered the demodulator reports it here (`on_frame_decoded`) and the classes
// pick their bytes up from its frame session.  Every demodulator has a session and an 8-frame history of its own (up to 8 demodulators
// per process; a further one is simply not batched), so the frames of several receivers in one process do not push each other out.
//
// A class may take a batcher result only when it is the result the class itself would compute:
//   * the soft bits it is handed are, byte for byte, the slice of a frame the batcher has decoded (compared with memcmp against the
//     batcher's host copy of that frame: the reference app passes frames through a ring buffer between its OFDM and radio threads,
//     src/../examples/app_helpers/app_ofdm_blocks.h:32-35, so pointers prove nothing), and
//   * for an MSC_Decoder, the 15 CIFs it was handed before were the 15 CIFs the batcher saw before that one (the time de-interleaver
//     reads 16 CIFs): the decoder counts how many consecutive batcher CIFs it has consumed.
// Otherwise -- a frame that did not come from this process's OFDM_Demod, a decoder that joined less than 16 CIFs ago, a skipped CIF --
// the class decodes on its own as before (its own 16-CIF ring is always kept up to date), with identical bytes.
// DABGPU_MIRROR_BATCH=0 in the environment switches the batcher off.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

#include "dabgpu.h"

namespace dabgpu_frame_batcher {

bool enabled();
// what the demodulators' receivers decode with every frame: the sub-channels registered, whether a FIC_Decoder listens; the value
// returned changes whenever the answer does
uint64_t subscription(std::vector<dabgpu_subchannel>& subs, bool& fic);
// OFDM_Demod `producer` (any address that identifies the demodulator): frame `gen` of its receiver's `session` (230400 soft bits) has been
// delivered, decoded for the subscription as it was when the frame was submitted; its destructor takes the demodulator out
void on_frame_decoded(const void* producer, dabgpu_frame_session* session, uint64_t gen, const int8_t* frame_bits);
void remove_producer(const void* producer);
// decoders: registration (reference-counted per distinct sub-channel) -- the FIC counts as a registration of its own
void add_fic();
void remove_fic();
void add_subchannel(const dabgpu_subchannel& sc);
void remove_subchannel(const dabgpu_subchannel& sc);

struct cif_id { int src = -1; uint64_t gen = ~0ull; int cif = -1; bool valid() const { return cif >= 0 && src >= 0; } };     // (demodulator, frame, CIF)
// which decoded frame (newest first) holds these 2304 soft bits as FIB group `group`?  On a hit the decoded bytes are copied out.
bool fetch_fib_group(const int8_t* group_bits, int group, uint8_t* bytes96, uint32_t* crc_mask, uint64_t* path_error);
// which (frame, CIF) holds these soft bits as the sub-channel's slice?  `after`: the CIF this decoder consumed last (tried first)
cif_id match_cif(const int8_t* slice_bits, size_t start_bit, size_t n_bits, cif_id after);
bool fetch_cif(cif_id id, const dabgpu_subchannel& sc, uint8_t* bytes, size_t capacity, size_t* n_bytes, uint64_t* path_error);
inline cif_id successor(cif_id a) { cif_id b; b.src = a.src; b.gen = a.cif == 3 ? a.gen + 1 : a.gen; b.cif = a.cif == 3 ? 0 : a.cif + 1; return b; }

// how the decoders of this process got their results so far (diagnostics: tests, DABGPU_MIRROR_PROFILE): picked up from a frame's batched
// decode, or decoded on their own, call by call (the first 16 CIFs of a decoder, frames submitted before it existed, a consumer that fell behind)
struct Counters { unsigned long long fib_groups_batched, fib_groups_call_by_call, cifs_batched, cifs_call_by_call; };
Counters counters();
void count_call_by_call(bool fib_group);
}  // namespace dabgpu_frame_batcher
