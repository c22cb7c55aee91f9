// dab/dabgpu_shared_context.h -- one device context shared by every decoder object of the process (the reference
// creates one DAB_Viterbi_Decoder per FIC_Decoder / MSC_Decoder; they share nothing but constant tables).
// Device chosen by DABGPU_DEVICE (default 0).  Throws std::runtime_error when no gfx950 device is usable.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct dabgpu_ctx;
dabgpu_ctx* dabgpu_shared_context();
int dabgpu_tie_rule_from_env();
