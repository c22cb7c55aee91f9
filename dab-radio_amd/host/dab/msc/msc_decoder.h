// dab/msc/msc_decoder.h -- MSC_Decoder with the reference's public interface (src/dab/msc/msc_decoder.h:16-38)
// over the MI355X C ABI.  In a receiver (an OFDM_Demod of this process feeds it) the decoder picks its bytes up from the demodulator's
// batched decode of the whole frame and owns NO device object; decoded call by call (bits from a file, the first frames after its
// creation) it creates, on that first call, a context of its own and a 16-CIF device ring: one launch per CIF does the time de-interleave
// (by index), de-puncture (by index), Viterbi and descrambling.
#pragma once
#include <stdint.h>
#include <vector>
#include "../database/dab_database_entities.h"
#include "utility/span.h"
#include "viterbi_config.h"

struct dabgpu_msc_stream;

class MSC_Decoder {
public:
    explicit MSC_Decoder(const Subchannel subchannel);
    ~MSC_Decoder();
    MSC_Decoder(const MSC_Decoder&) = delete;
    MSC_Decoder& operator=(const MSC_Decoder&) = delete;
    // returns a view of the decoded bytes, empty until 16 CIFs have been seen (or on a size error)
    tcb::span<uint8_t> DecodeCIF(tcb::span<const viterbi_bit_t> buf);
    uint64_t GetLastPathError() const { return m_last_error; }

private:
    const Subchannel m_subchannel;
    void EnsureStream();
    dabgpu_msc_stream* m_stream;       // null until the first call-by-call decode
    struct dabgpu_ctx* m_ctx;          // then: this decoder's own device context (stream + scratch): decoders of different threads run side by side
    std::vector<uint8_t> m_decoded_bytes;
    uint64_t m_last_error = 0;
    // frame batcher (dab/dabgpu_frame_batcher.h): the sub-channel as registered, the batcher CIF consumed last, how many in a row
    struct BatchState;
    BatchState* m_batch;
};
