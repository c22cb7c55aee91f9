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
int dabgpu_tie_rule_from_env();
