// receiver_bank.h -- internal interface between receiver.hip (the dabgpu_receiver_* entry points) and receiver_bank.hip
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "dabgpu.h"

#define DABGPU_RX_BANK_MAX 64           // receivers of one process on one device that share the bank

struct dabgpu_frame_session;
struct dabgpu_rx_bank;
struct dabgpu_rx_member;

// h_stage: the receiver's three page-locked staging buffers (they stay the receiver's)
int dabgpu_rx_bank_join(int device, float* const* h_stage, dabgpu_rx_member** out);
void dabgpu_rx_bank_leave(dabgpu_rx_member* m);                 // waits for the member's jobs
dabgpu_frame_session* dabgpu_rx_bank_session(dabgpu_rx_member* m);
int dabgpu_rx_bank_set_subchannels(dabgpu_rx_member* m, const dabgpu_subchannel* subs, int n, int decode_fic);
int dabgpu_rx_bank_reset(dabgpu_rx_member* m);
int dabgpu_rx_bank_post_sync(dabgpu_rx_member* m, const dabgpu_sync_cfg* cfg, int stage, size_t prs_sample);
int dabgpu_rx_bank_sync_pending(dabgpu_rx_member* m);
int dabgpu_rx_bank_wait_sync(dabgpu_rx_member* m, dabgpu_sync_state* out, float* h_impulse, float* h_freq_response);
int dabgpu_rx_bank_post_frame(dabgpu_rx_member* m, int stage, size_t frame_sample, float beta, int want_views, int tie, uint64_t* generation);
int dabgpu_rx_bank_wait_stage(dabgpu_rx_member* m, int stage);
int dabgpu_rx_bank_wait_frame(dabgpu_rx_member* m, uint64_t generation, dabgpu_receiver_frame* out);
void dabgpu_rx_bank_shutdown(void);
