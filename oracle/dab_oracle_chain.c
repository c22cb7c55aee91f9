/*
 * dab_oracle_chain.c -- CPU ORACLE (test infrastructure, NOT product code): the oracle's per-frame functions composed into the
 * sequence one receiver runs per transmission frame, as ONE C call so that it can be timed on many host threads without the
 * interpreter in between (bench.py `cpu_baseline_full`) and compared with the GPU's chained path (tests/).
 *
 * Order per frame = SURVEY.md A.2 steps 2-10 followed by BasicRadio's fan-out:
 *   RunCoarseFreqSync + RunFineTimeSync        src/ofdm/ofdm_demodulator.cpp:360-548
 *   PipelineThread (PLL, CP phase, FFT, DQPSK) src/ofdm/ofdm_demodulator.cpp:650-766
 *   fine-frequency update                      src/ofdm/ofdm_demodulator.cpp:606-618
 *   FIC: 4 x FIC_Decoder::DecodeFIBGroup       src/basic_radio/basic_fic_runner.cpp:34-49, src/dab/fic/fic_decoder.cpp:53-117
 *   MSC: per CIF and sub-channel CIF_Deinterleaver + MSC_Decoder::DecodeCIF
 *                                              src/basic_radio/basic_radio.cpp:41-65, src/dab/msc/msc_decoder.cpp:46-154
 */
#include <stdlib.h>
#include <string.h>

#include "dab_oracle.h"

int dab_receive_frames(const dab_cf32 *slices, size_t n_distinct, size_t stride, size_t prs_offset, size_t n_total,
                       const dab_subchannel *subs, int n_subs, int tie_rule, dab_sync_state *state,
                       uint32_t *n_fib_crc_ok, uint32_t *n_sync_failed, uint8_t *fib_last /*[4][96]*/,
                       uint8_t *msc_last /*[4][sum of decoded bytes]*/, uint64_t *digest)
{
    if (!slices || n_distinct == 0 || !state || n_subs < 0 || n_subs > 64 || prs_offset < DAB_NB_CYCLIC_PREFIX ||
        stride < prs_offset + (DAB_NB_FFT - DAB_NB_CYCLIC_PREFIX) + DAB_NB_FRAME_SAMPLES) return -1;
    int mapper[DAB_NB_DATA_CARRIERS];
    dab_get_mapper(mapper);
    dab_cf32 *prs = malloc(sizeof(dab_cf32) * DAB_NB_FFT), *conj_ref = malloc(sizeof(dab_cf32) * DAB_NB_FFT),
             *time_ref = malloc(sizeof(dab_cf32) * DAB_NB_FFT);
    int8_t *bits = malloc(DAB_NB_FRAME_BITS), *logical = malloc(DAB_NB_CIF_BITS);
    dab_deinterleaver *deint[64];
    int out_off[65], nbytes[64];
    dab_get_prs_fft(prs);
    dab_sync_refs(prs, conj_ref, time_ref);
    dab_sync_cfg cfg;
    dab_sync_cfg_default(&cfg);
    out_off[0] = 0;
    for (int s = 0; s < n_subs; s++) {
        int pi[4], lx[4];
        if (dab_subchannel_plan(&subs[s], pi, lx, &nbytes[s]) < 0) return -1;
        deint[s] = dab_deinterleaver_create(subs[s].length * 8);
        out_off[s + 1] = out_off[s] + nbytes[s];
    }
    uint8_t *msc = malloc((size_t)4 * (out_off[n_subs] > 0 ? out_off[n_subs] : 1)), fib[4][96];
    uint32_t ok_fibs = 0, bad_sync = 0;
    uint64_t dg = 0;
    for (size_t k = 0; k < n_total; k++) {
        const dab_cf32 *slice = slices + (k % n_distinct) * stride;
        const dab_cf32 *prs_sym = slice + prs_offset;
        dab_coarse_freq_sync(prs_sym, time_ref, &cfg, state, NULL);
        const float f = state->freq_coarse + state->freq_fine;                      /* :480, :672 */
        int off = 0;
        if (!dab_fine_time_sync(prs_sym, conj_ref, &cfg, f, &off, NULL)) {           /* :529-532: Reset() */
            bad_sync++;
            state->freq_coarse = 0.0f; state->freq_fine = 0.0f; state->is_found_coarse = 0; state->total_frames_desync++;
            continue;
        }
        state->fine_time_offset = off;
        const float total = dab_demod_frame(prs_sym + off, f, mapper, bits, NULL, NULL, NULL);
        state->freq_fine = dab_update_fine_freq(state->freq_fine, total);
        state->total_frames_read++;
        for (int g = 0; g < 4; g++) {
            uint32_t mask = 0;
            dg += dab_fic_decode_group(bits + g * DAB_NB_FIB_GROUP_BITS, tie_rule, fib[g], &mask);
            ok_fibs += (mask & 1u) + ((mask >> 1) & 1u) + ((mask >> 2) & 1u);
        }
        for (int c = 0; c < DAB_NB_CIFS; c++) {
            const int8_t *cif = bits + DAB_NB_FIC_BITS + (size_t)c * DAB_NB_CIF_BITS;
            for (int s = 0; s < n_subs; s++) {
                uint8_t *dst = msc + (size_t)c * out_off[n_subs] + out_off[s];
                dab_deinterleaver_consume(deint[s], cif + subs[s].start_address * 64);
                if (!dab_deinterleaver_deinterleave(deint[s], logical)) { memset(dst, 0, (size_t)nbytes[s]); continue; }
                int nb = 0;
                dg += dab_msc_decode_logical(&subs[s], logical, tie_rule, dst, &nb);
                for (int i = 0; i < nb; i += 8) dg += dst[i];
            }
        }
    }
    if (fib_last) memcpy(fib_last, fib, sizeof(fib));
    if (msc_last && out_off[n_subs] > 0) memcpy(msc_last, msc, (size_t)4 * out_off[n_subs]);
    if (n_fib_crc_ok) *n_fib_crc_ok = ok_fibs;
    if (n_sync_failed) *n_sync_failed = bad_sync;
    if (digest) *digest = dg;
    for (int s = 0; s < n_subs; s++) dab_deinterleaver_destroy(deint[s]);
    free(msc); free(bits); free(logical); free(prs); free(conj_ref); free(time_ref);
    return 0;
}
