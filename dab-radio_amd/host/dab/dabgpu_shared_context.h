// dab/dabgpu_shared_context.h -- one device context shared by every decoder object of the process (the reference
// creates one DAB_Viterbi_Decoder per FIC_Decoder / MSC_Decoder; they share nothing but constant tables).
// Device chosen by DABGPU_DEVICE (default 0).  Throws std::runtime_error when no gfx950 device is usable.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct dabgpu_ctx;
dabgpu_ctx* dabgpu_shared_context();
// A context of its own (stream + scratch) on the same device, destroyed by the caller with dabgpu_destroy: MSC_Decoder takes one so
// that the reference's thread pool -- one task per sub-channel, src/basic_radio/basic_radio.cpp:51-62 -- decodes its sub-channels
// side by side on the device instead of queueing on the shared context's lock.
dabgpu_ctx* dabgpu_private_context();
// Which upstream Viterbi core the decoders model (the `tie_rule` argument of the C ABI): 0 = ViterbiDecoder_Scalar (wrapping uint16_t sums,
// strict compare), 1 = the SIMD cores ViterbiDecoder_AVX_u16 / _SSE_u16 / _NEON_u16 (saturating sums, min + compare-equal).
// DABGPU_VITERBI_CORE=scalar|simd decides (DABGPU_TIE_RULE=0|1 is the older name of the same switch); without either, the model is the
// core the reference's own build would have selected on THIS host (src/dab/algorithms/dab_viterbi_decoder.cpp:51-73 under its default
// -march=native preset): AVX2 or SSE4.1 present, or AArch64 -> simd; otherwise scalar.
int dabgpu_core_model_from_env();
inline int dabgpu_tie_rule_from_env() { return dabgpu_core_model_from_env(); }
