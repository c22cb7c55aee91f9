"""dabgpu -- thin ctypes binding over the C ABI in include/dabgpu.h (dab-radio_amd/libdabgpu.so).

This is plumbing for tests and bench.py: device memory comes from torch tensors (or any object with
`data_ptr()`), the compute is the hand-written HIP in dab-radio_amd/csrc.  There is NO CPU fallback:
if the shared library is missing or no gfx950 device is present, calls raise DabGpuError.
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("DABGPU_LIB") or os.path.join(_ROOT, "libdabgpu.so")   # DABGPU_LIB: development A/B builds

NB_FRAME_SYMBOLS = 76
NB_SYMBOL_PERIOD = 2552
NB_NULL_PERIOD = 2656
NB_FFT = 2048
NB_CP = 504
NB_CARRIERS = 1536
NB_FRAME_SAMPLES = 196608
NB_SYM_BITS = 3072
NB_FRAME_BITS = 230400
NB_FIC_BITS = 9216
NB_FIB_GROUP_BITS = 2304
NB_CIF_BITS = 55296
ABI_VERSION = 4                          # DABGPU_ABI_VERSION of include/dabgpu.h
BITS_NATURAL, BITS_MSC_CLASSED = 0, 1     # dabgpu_ofdm_demod_frames_history / dabgpu_msc_decode_frames_layout


def classed_to_natural_index():
    """index array P with natural_frame_bits == classed_frame_bits[P]: FIC unchanged, inside each CIF row bit i sits at
    (i mod 16) * 3456 + i // 16 (DABGPU_BITS_MSC_CLASSED)"""
    import numpy as np
    i = np.arange(NB_CIF_BITS)
    row = (i % 16) * (NB_CIF_BITS // 16) + i // 16
    return np.concatenate([np.arange(NB_FIC_BITS)] + [NB_FIC_BITS + q * NB_CIF_BITS + row for q in range(4)])

# every symbol include/dabgpu.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "dabgpu_strerror", "dabgpu_last_error", "dabgpu_abi_version", "dabgpu_device_count",
    "dabgpu_create", "dabgpu_destroy", "dabgpu_synchronize", "dabgpu_host_pin", "dabgpu_host_unpin",
    "dabgpu_get_prs_fft_ref", "dabgpu_get_carrier_mapper", "dabgpu_get_fft_twiddles",
    "dabgpu_ofdm_demod_frames", "dabgpu_ofdm_phase_update", "dabgpu_ofdm_demod_frames_host_sync", "dabgpu_ofdm_demod_stream_frame_sync",
    "dabgpu_sync_cfg_default", "dabgpu_ofdm_sync", "dabgpu_ofdm_sync_host_sync",
    "dabgpu_viterbi_set_mapping", "dabgpu_ofdm_auto_symbols_per_block", "dabgpu_viterbi_decode_batch", "dabgpu_fic_decode_frames", "dabgpu_subchannel_plan", "dabgpu_subchannel_validate", "dabgpu_msc_decode_frames",
    "dabgpu_fic_decode_group_host_sync", "dabgpu_viterbi_decode_host_sync", "dabgpu_msc_stream_create",
    "dabgpu_msc_stream_destroy", "dabgpu_msc_stream_push_cif", "dabgpu_msc_stream_deinterleave_sync",
    "dabgpu_msc_stream_decode_sync",
    "dabgpu_iq_format_from_mode", "dabgpu_iq_format_sample_bytes", "dabgpu_wav_parse_header",
    "dabgpu_iq_convert", "dabgpu_iq_convert_host_sync", "dabgpu_ofdm_demod_frames_raw",
    "dabgpu_soft_bits_to_hard_bytes", "dabgpu_hard_bytes_to_soft_bits",
    "dabgpu_soft_bits_to_hard_bytes_host_sync", "dabgpu_hard_bytes_to_soft_bits_host_sync",
    "dabgpu_stream_cfg_default", "dabgpu_stream_bank_create", "dabgpu_stream_bank_create_mode", "dabgpu_stream_bank_destroy", "dabgpu_stream_bank_reset",
    "dabgpu_stream_bank_process", "dabgpu_stream_bank_process_raw", "dabgpu_stream_bank_status",
    "dabgpu_stream_bank_process_retained", "dabgpu_stream_bank_release",
    "dabgpu_dabplus_bank_create", "dabgpu_dabplus_bank_destroy", "dabgpu_dabplus_bank_reset", "dabgpu_dabplus_bank_process",
    "dabgpu_dabplus_process_frame_host_sync",
    "dabgpu_get_ofdm_params", "dabgpu_ofdm_demod_frames_mode", "dabgpu_ofdm_phase_update_mode",
    "dabgpu_ofdm_sync_mode", "dabgpu_ofdm_demod_stream_frame_sync_mode", "dabgpu_ofdm_sync_host_sync_mode",
    "dabgpu_stream_bank_process_ring", "dabgpu_fic_decode_ring", "dabgpu_msc_decode_ring", "dabgpu_dabplus_bank_process_masked",
    "dabgpu_ofdm_demod_frames_history", "dabgpu_msc_decode_frames_layout", "dabgpu_stream_bank_process_ring_layout",
    "dabgpu_msc_decode_ring_layout", "dabgpu_ofdm_demod_phase_frames", "dabgpu_decode_frames_layout", "dabgpu_decode_ring_layout",
    "dabgpu_frame_session_create", "dabgpu_frame_session_destroy", "dabgpu_frame_session_set_subchannels", "dabgpu_frame_session_push_frame",
    "dabgpu_frame_session_fetch_fib_group", "dabgpu_frame_session_fetch_cif",
    "dabgpu_viterbi_decode_depunctured_host_sync", "dabgpu_stream_bank_process_ring_retained",
    "dabgpu_ofdm_tune", "dabgpu_ofdm_tuned_symbols_per_block", "dabgpu_ofdm_sync_demod_frames",
    "dabgpu_multiplex_mapping",
    "dabgpu_receiver_create", "dabgpu_receiver_create_banked", "dabgpu_receiver_destroy", "dabgpu_receiver_session", "dabgpu_receiver_set_subchannels", "dabgpu_receiver_stage",
    "dabgpu_receiver_reset", "dabgpu_receiver_submit_sync", "dabgpu_receiver_wait_sync", "dabgpu_receiver_submit_frame", "dabgpu_receiver_wait_frame",
    "dabgpu_receiver_submit_demod", "dabgpu_receiver_submit_decode",
    "dabgpu_ingest_create", "dabgpu_ingest_destroy", "dabgpu_ingest_acquire", "dabgpu_ingest_submit", "dabgpu_ingest_wait", "dabgpu_ingest_consumed",
]

IQ_FORMATS = ["raw_u8", "raw_s8", "raw_s16l", "raw_s16b", "raw_u16l", "raw_u16b", "raw_s32l", "raw_s32b", "raw_u32l", "raw_u32b",
              "raw_f32l", "raw_f32b", "raw_f64l", "raw_f64b",
              "wav_pcm8", "wav_pcm16", "wav_pcm24", "wav_pcm32", "wav_f32", "wav_f64", "wav_alaw", "wav_mulaw"]


class WavHeader(C.Structure):
    """dabgpu_wav_header"""
    _fields_ = [("iq_format", C.c_int32), ("audio_format", C.c_uint16), ("total_channels", C.c_uint16),
                ("samples_per_second", C.c_uint32), ("average_bytes_per_second", C.c_uint32),
                ("data_block_align_bytes", C.c_uint16), ("bits_per_sample", C.c_uint16),
                ("data_chunk_size", C.c_uint32), ("data_chunk_offset", C.c_uint64)]


class Codeword(C.Structure):
    """dabgpu_codeword (include/dabgpu.h)"""
    _fields_ = [("d_src", C.c_uint64), ("d_out", C.c_uint64), ("n_steps", C.c_uint32),
                ("seg_pi", C.c_uint32 * 4), ("seg_steps", C.c_uint32 * 4), ("start_state", C.c_uint32),
                ("n_crc_blocks", C.c_uint32), ("n_slots", C.c_uint32), ("newest_slot", C.c_uint32),
                ("cifs_per_frame", C.c_uint32), ("frame_stride", C.c_uint32), ("cif_stride", C.c_uint32),
                ("end_state", C.c_uint32), ("flags", C.c_uint32)]


class CodewordResult(C.Structure):
    """dabgpu_codeword_result"""
    _fields_ = [("path_error", C.c_uint64), ("crc_ok_mask", C.c_uint32), ("n_out_bytes", C.c_uint32)]


class SubChannel(C.Structure):
    """dabgpu_subchannel"""
    _fields_ = [("start_address", C.c_int), ("length", C.c_int), ("is_uep", C.c_int),
                ("uep_prot_index", C.c_int), ("eep_prot_level", C.c_int), ("eep_type", C.c_int)]


class SyncCfg(C.Structure):
    """dabgpu_sync_cfg"""
    _fields_ = [("fine_freq_update_beta", C.c_float), ("is_coarse_freq_correction", C.c_int),
                ("max_coarse_freq_correction_norm", C.c_float), ("coarse_freq_slow_beta", C.c_float),
                ("impulse_peak_threshold_db", C.c_float), ("impulse_peak_distance_probability", C.c_float)]


class SyncState(C.Structure):
    """dabgpu_sync_state"""
    _fields_ = [("freq_coarse", C.c_float), ("freq_fine", C.c_float), ("is_found_coarse", C.c_int),
                ("fine_time_offset", C.c_int), ("sync_valid", C.c_int), ("reserved", C.c_int)]


class StreamCfg(C.Structure):
    """dabgpu_stream_cfg"""
    _fields_ = [("signal_l1_update_beta", C.c_float), ("signal_l1_nb_samples", C.c_int), ("signal_l1_nb_decimate", C.c_int),
                ("thresh_null_start", C.c_float), ("thresh_null_end", C.c_float), ("sync", SyncCfg)]


STREAM_STATUS_DTYPE = [("state", "<i4"), ("signal_l1_average", "<f4"), ("freq_coarse", "<f4"), ("freq_fine", "<f4"),
                       ("is_found_coarse", "<i4"), ("fine_time_offset", "<i4"), ("total_frames_read", "<i4"),
                       ("total_frames_desync", "<i4")]
SUPERFRAME_RESULT_DTYPE = [("rs_failed_index", "<i4"), ("rs_corrected", "<i4"), ("firecode_ok", "<i4"), ("header_valid", "<i4"),
                           ("descriptor", "<i4"), ("num_aus", "<i4"), ("au_start", "<i4", (8,)), ("au_walk_stopped_at", "<i4"),
                           ("au_crc_ok_mask", "<u4"), ("frame_index", "<i4"), ("firecode_rx_calc", "<u4"),
                           ("au_crc_calc", "<u2", (6,)), ("reserved", "<u2", (2,))]
SYNC_STATE_DTYPE = [("freq_coarse", "<f4"), ("freq_fine", "<f4"), ("is_found_coarse", "<i4"),
                    ("fine_time_offset", "<i4"), ("sync_valid", "<i4"), ("reserved", "<i4")]
RESULT_DTYPE = [("path_error", "<u8"), ("crc_ok_mask", "<u4"), ("n_out_bytes", "<u4")]


class DabGpuError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libdabgpu.so; loud failure when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DabGpuError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dab-radio_amd/csrc` (no CPU fallback exists)")
        # torch bundles its own libamdhip64.so.7; when torch is in the process it must be the one HIP runtime
        # (same SONAME as /opt/rocm's): import it first so libdabgpu.so binds to the already-loaded copy and
        # tensors, streams and our kernels share one runtime.  Stand-alone C/C++ users link /opt/rocm's.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        if L.dabgpu_abi_version() != ABI_VERSION:
            raise DabGpuError(f"{LIB_PATH} implements ABI version {L.dabgpu_abi_version()}, this binding was written for {ABI_VERSION}: rebuild it")
        L.dabgpu_strerror.restype = C.c_char_p
        L.dabgpu_strerror.argtypes = [C.c_int]
        L.dabgpu_last_error.restype = C.c_char_p
        L.dabgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]
        L.dabgpu_destroy.argtypes = [C.c_void_p]
        L.dabgpu_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        L.dabgpu_get_prs_fft_ref.argtypes = [C.c_int, C.c_void_p]
        L.dabgpu_get_carrier_mapper.argtypes = [C.c_int, C.c_void_p]
        L.dabgpu_get_fft_twiddles.argtypes = [C.c_void_p]
        L.dabgpu_ofdm_demod_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        L.dabgpu_ofdm_demod_frames_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        L.dabgpu_ofdm_demod_frames_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.dabgpu_ofdm_demod_phase_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_msc_decode_frames_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dabgpu_decode_frames_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dabgpu_decode_ring_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dabgpu_ofdm_demod_stream_frame_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_phase_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_demod_frames_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_sync_cfg_default.argtypes = [C.c_void_p]
        L.dabgpu_ofdm_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_sync_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_viterbi_set_mapping.argtypes = [C.c_void_p, C.c_int]
        L.dabgpu_ofdm_auto_symbols_per_block.argtypes = [C.c_void_p, C.c_size_t]
        L.dabgpu_ofdm_tune.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_tuned_symbols_per_block.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int]
        L.dabgpu_ofdm_sync_demod_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.dabgpu_viterbi_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.dabgpu_fic_decode_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p]
        L.dabgpu_subchannel_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_msc_decode_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_iq_format_from_mode.argtypes = [C.c_char_p]
        L.dabgpu_iq_format_sample_bytes.restype = C.c_size_t
        L.dabgpu_iq_format_sample_bytes.argtypes = [C.c_int]
        L.dabgpu_wav_parse_header.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.dabgpu_iq_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.dabgpu_iq_convert_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        L.dabgpu_soft_bits_to_hard_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.dabgpu_hard_bytes_to_soft_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.dabgpu_soft_bits_to_hard_bytes_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dabgpu_hard_bytes_to_soft_bits_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dabgpu_stream_cfg_default.argtypes = [C.c_void_p]
        L.dabgpu_stream_bank_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
        L.dabgpu_stream_bank_create_mode.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
        L.dabgpu_stream_bank_destroy.argtypes = [C.c_void_p]
        L.dabgpu_stream_bank_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.dabgpu_stream_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_void_p]
        L.dabgpu_stream_bank_process_retained.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                          C.c_void_p, C.c_void_p]
        L.dabgpu_stream_bank_release.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        L.dabgpu_stream_bank_process_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                     C.c_void_p, C.c_void_p]
        L.dabgpu_stream_bank_process_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_void_p]
        L.dabgpu_stream_bank_process_ring_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int,
                                                             C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_stream_bank_process_ring_retained.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                                               C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_msc_decode_ring_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dabgpu_fic_decode_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p]
        L.dabgpu_msc_decode_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_dabplus_bank_process_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                                         C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_stream_bank_status.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_dabplus_bank_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.dabgpu_dabplus_bank_destroy.argtypes = [C.c_void_p]
        L.dabgpu_dabplus_bank_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.dabgpu_dabplus_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.dabgpu_dabplus_process_frame_host_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                             C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_sync_mode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.dabgpu_ofdm_sync_host_sync_mode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_get_ofdm_params.argtypes = [C.c_int, C.c_void_p]
        L.dabgpu_ofdm_demod_frames_mode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.c_void_p]
        L.dabgpu_ofdm_phase_update_mode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
        L.dabgpu_ingest_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.dabgpu_ingest_destroy.argtypes = [C.c_void_p]
        L.dabgpu_ingest_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.dabgpu_ingest_submit.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.dabgpu_ingest_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dabgpu_ingest_consumed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        L = lib()
        raise DabGpuError(f"{what}: {L.dabgpu_strerror(status).decode()} -- {L.dabgpu_last_error().decode()}")


def _ptr(x):
    """device/host pointer of a torch tensor, numpy array, int or None"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return C.c_void_p(x.ctypes.data)
    raise TypeError(f"cannot take a pointer of {type(x)}")


def device_count():
    return lib().dabgpu_device_count()


def host_tables():
    """(prs_fft_ref complex64[2048], carrier_mapper int32[1536], twiddles complex64[2048]) from the product's host code"""
    import numpy as np
    prs = np.zeros(NB_FFT, dtype=np.complex64)
    mapper = np.zeros(NB_CARRIERS, dtype=np.int32)
    tw = np.zeros(NB_FFT, dtype=np.complex64)
    check(lib().dabgpu_get_prs_fft_ref(1, _ptr(prs)), "get_prs_fft_ref")
    check(lib().dabgpu_get_carrier_mapper(1, _ptr(mapper)), "get_carrier_mapper")
    check(lib().dabgpu_get_fft_twiddles(_ptr(tw)), "get_fft_twiddles")
    return prs, mapper, tw


class Context:
    """One device + constant tables (dabgpu_create / dabgpu_destroy)."""

    def __init__(self, device=0, prs_fft_ref=None, carrier_mapper=None):
        import numpy as np
        self._h = C.c_void_p()
        prs = None if prs_fft_ref is None else np.ascontiguousarray(prs_fft_ref, dtype=np.complex64)
        mp = None if carrier_mapper is None else np.ascontiguousarray(carrier_mapper, dtype=np.int32)
        check(lib().dabgpu_create(C.byref(self._h), device, _ptr(prs), _ptr(mp)), "dabgpu_create")
        self.device = device

    def close(self):
        if self._h:
            lib().dabgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream(stream):
        if stream is None:
            try:
                import torch
                if torch.cuda.is_available():
                    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
            except Exception:
                pass
            return None
        return C.c_void_p(int(stream))

    def synchronize(self, stream=None):
        check(lib().dabgpu_synchronize(self._h, self._stream(stream)), "dabgpu_synchronize")

    def ofdm_demod_frames(self, iq, bits, freq_offset=None, cp_corr=None, fft=None, symbols_per_block=0,
                          n_frames=None, stream=None, dqpsk=None, bits_frame_stride=0):
        """Launch the fused PLL+CP-phase+FFT+DQPSK+demap kernel on device buffers (asynchronous)."""
        if n_frames is None:
            n_frames = iq.numel() // NB_FRAME_SAMPLES if hasattr(iq, "numel") else None
        check(lib().dabgpu_ofdm_demod_frames(self._h, _ptr(iq), n_frames, _ptr(freq_offset), _ptr(bits),
                                             _ptr(cp_corr), _ptr(fft), _ptr(dqpsk), symbols_per_block, bits_frame_stride, self._stream(stream)),
              "dabgpu_ofdm_demod_frames")

    def ofdm_demod_frames_raw(self, raw, fmt, n_frames, bits, freq_offset=None, cp_corr=None, fft=None, symbols_per_block=0,
                              stream=None, dqpsk=None, bits_frame_stride=0):
        """The same from frames still in capture format number `fmt` (u8 / s8 / s16l are read by the kernel itself)."""
        check(lib().dabgpu_ofdm_demod_frames_raw(self._h, _ptr(raw), int(fmt), n_frames, _ptr(freq_offset), _ptr(bits),
                                                 _ptr(cp_corr), _ptr(fft), _ptr(dqpsk), symbols_per_block, bits_frame_stride,
                                                 self._stream(stream)), "dabgpu_ofdm_demod_frames_raw")

    def ofdm_demod_frames_history(self, raw, fmt, n_frames, bits, freq_offset=None, cp_corr=None, symbols_per_block=0,
                                  bits_frame_stride=0, bits_layout=BITS_NATURAL, stream=None):
        """Soft bits straight into the decoder's frame-history ring; bits_layout = BITS_MSC_CLASSED stores the MSC part in
        time-interleaver class order (read it with msc_decode_frames(..., bits_layout=BITS_MSC_CLASSED))."""
        check(lib().dabgpu_ofdm_demod_frames_history(self._h, _ptr(raw), int(fmt), n_frames, _ptr(freq_offset), _ptr(bits), _ptr(cp_corr),
                                                     symbols_per_block, bits_frame_stride, int(bits_layout), self._stream(stream)),
              "dabgpu_ofdm_demod_frames_history")

    def ofdm_demod_phase_frames(self, raw, fmt, n_frames, bits, freq_offset=None, cp_corr=None, symbols_per_block=0, bits_frame_stride=0,
                                bits_layout=BITS_NATURAL, beta=0.0, total_phase=None, fine_freq=None, stream=None):
        """demodulation + phase tail (ofdm_phase_update) as one call; one launch when a workgroup walks a whole frame"""
        check(lib().dabgpu_ofdm_demod_phase_frames(self._h, _ptr(raw), int(fmt), n_frames, _ptr(freq_offset), _ptr(bits), _ptr(cp_corr),
                                                   symbols_per_block, bits_frame_stride, int(bits_layout), float(beta), _ptr(total_phase),
                                                   _ptr(fine_freq), self._stream(stream)), "dabgpu_ofdm_demod_phase_frames")

    def ofdm_demod_frames_mode(self, mode, iq, n_frames, bits, freq_offset=None, cp_corr=None, fft=None, symbols_per_block=0, stream=None):
        """frame-aligned frames of transmission mode 1..4 through the size-generic kernel"""
        check(lib().dabgpu_ofdm_demod_frames_mode(self._h, int(mode), _ptr(iq), n_frames, _ptr(freq_offset), _ptr(bits), _ptr(cp_corr),
                                                  _ptr(fft), symbols_per_block, self._stream(stream)), "dabgpu_ofdm_demod_frames_mode")

    def ofdm_sync_host_mode(self, mode, prs_sym, state, cfg=None):
        """numpy convenience over dabgpu_ofdm_sync_host_sync_mode: returns (impulse_db[nb_fft], freq_response_db[nb_fft])"""
        import numpy as np
        n = ofdm_params(mode)["nb_fft"]
        cfg = cfg or sync_cfg_default()
        sym = np.ascontiguousarray(prs_sym, dtype=np.complex64).reshape(-1)
        assert sym.size == n
        imp = np.empty(n, dtype=np.float32)
        frq = np.empty(n, dtype=np.float32)
        check(lib().dabgpu_ofdm_sync_host_sync_mode(self._h, int(mode), _ptr(sym), C.byref(cfg), C.byref(state), _ptr(imp), _ptr(frq)),
              "dabgpu_ofdm_sync_host_sync_mode")
        return imp, frq

    def ofdm_phase_update_mode(self, mode, cp_corr, n_frames, total_phase=None, fine_freq=None, beta=0.9, stream=None):
        check(lib().dabgpu_ofdm_phase_update_mode(self._h, int(mode), _ptr(cp_corr), n_frames, beta, _ptr(total_phase), _ptr(fine_freq),
                                                  self._stream(stream)), "dabgpu_ofdm_phase_update_mode")

    def ofdm_phase_update(self, cp_corr, n_frames, total_phase=None, fine_freq=None, beta=0.9, stream=None):
        check(lib().dabgpu_ofdm_phase_update(self._h, _ptr(cp_corr), n_frames, beta, _ptr(total_phase),
                                             _ptr(fine_freq), self._stream(stream)), "dabgpu_ofdm_phase_update")

    def ofdm_demod_frames_host(self, iq, freq_offset=None, want_fft=False):
        """numpy in / numpy out convenience over dabgpu_ofdm_demod_frames_host_sync"""
        import numpy as np
        iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(-1)
        n = iq.size // NB_FRAME_SAMPLES
        assert n * NB_FRAME_SAMPLES == iq.size
        f = None if freq_offset is None else np.ascontiguousarray(freq_offset, dtype=np.float32).reshape(-1)
        bits = np.empty((n, NB_FRAME_BITS), dtype=np.int8)
        total = np.empty(n, dtype=np.float32)
        fft = np.empty((n, 77, NB_FFT), dtype=np.complex64) if want_fft else None
        check(lib().dabgpu_ofdm_demod_frames_host_sync(self._h, _ptr(iq), n, _ptr(f), _ptr(bits), _ptr(total),
                                                       _ptr(fft)), "dabgpu_ofdm_demod_frames_host_sync")
        return bits, total, fft

    # ---- sync ----
    def ofdm_sync(self, prs_syms, n_streams, stride_samples, states, cfg=None, impulse=None, freq_response=None, stream=None):
        cfg = cfg or sync_cfg_default()
        check(lib().dabgpu_ofdm_sync(self._h, _ptr(prs_syms), n_streams, stride_samples, C.byref(cfg), _ptr(states),
                                     _ptr(impulse), _ptr(freq_response), self._stream(stream)), "dabgpu_ofdm_sync")

    def ofdm_sync_host(self, prs_sym, state, cfg=None):
        """numpy convenience: returns (state, impulse_response, freq_response)"""
        import numpy as np
        cfg = cfg or sync_cfg_default()
        x = np.ascontiguousarray(prs_sym, dtype=np.complex64)[:NB_FFT].copy()
        imp = np.empty(NB_FFT, dtype=np.float32)
        frq = np.empty(NB_FFT, dtype=np.float32)
        check(lib().dabgpu_ofdm_sync_host_sync(self._h, _ptr(x), C.byref(cfg), C.byref(state), _ptr(imp), _ptr(frq)),
              "dabgpu_ofdm_sync_host_sync")
        return state, imp, frq

    def ofdm_sync_demod_frames(self, iq, n_streams, stream_stride_samples, prs_offset_samples, states, bits, cfg=None, cp_corr=None,
                               symbols_per_block=0, bits_frame_stride=0, bits_layout=BITS_NATURAL, total_phase=None, stream=None):
        """PRS synchronisation -> demodulation at the position / with the offset it found -> fine-frequency update, one call (device
        buffers, asynchronous); `states` = [n_streams] dabgpu_sync_state records on the device, in/out"""
        cfg = cfg or sync_cfg_default()
        check(lib().dabgpu_ofdm_sync_demod_frames(self._h, _ptr(iq), n_streams, stream_stride_samples, prs_offset_samples, C.byref(cfg),
                                                  _ptr(states), _ptr(bits), _ptr(cp_corr), symbols_per_block, bits_frame_stride,
                                                  int(bits_layout), _ptr(total_phase), self._stream(stream)), "dabgpu_ofdm_sync_demod_frames")

    def ofdm_tune(self, raw, fmt, n_frames, bits, bits_frame_stride=0, bits_layout=BITS_NATURAL, with_phase_tail=False, stream=None):
        """explicit (blocking) calibration of symbols_per_block = 0 for this call shape; returns the run length recorded"""
        chosen = C.c_int(0)
        check(lib().dabgpu_ofdm_tune(self._h, _ptr(raw), int(fmt), n_frames, _ptr(bits), bits_frame_stride, int(bits_layout),
                                     int(bool(with_phase_tail)), self._stream(stream), C.byref(chosen)), "dabgpu_ofdm_tune")
        return chosen.value

    def ofdm_tuned_symbols_per_block(self, fmt, n_frames, bits_layout=BITS_NATURAL, with_phase_tail=False):
        """what symbols_per_block = 0 resolves to for that call shape right now"""
        return int(lib().dabgpu_ofdm_tuned_symbols_per_block(self._h, int(fmt), int(n_frames), int(bits_layout), int(bool(with_phase_tail))))

    # ---- channel decode ----
    def ofdm_auto_symbols_per_block(self, n_frames):
        """the run length dabgpu_ofdm_tune last recorded for the size bucket of n_frames (0 = nothing recorded)"""
        return int(lib().dabgpu_ofdm_auto_symbols_per_block(self._h, int(n_frames)))

    def viterbi_set_mapping(self, mapping):
        """0 = auto, 1 = one wavefront per codeword, 2 = one lane per codeword, 3 = eight lanes per codeword (include/dabgpu.h DABGPU_VIT_MAP_*)."""
        check(lib().dabgpu_viterbi_set_mapping(self._h, int(mapping)), "dabgpu_viterbi_set_mapping")

    def multiplex_mapping(self, n_ensembles, subchannels):
        """(mapping the MSC decode of this multiplex takes: 1 wave / 2 lane / 3 octet, modelled microseconds of the three)"""
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels)
        m, us = C.c_int(0), (C.c_double * 3)()
        lib().dabgpu_multiplex_mapping.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        check(lib().dabgpu_multiplex_mapping(self._h, n_ensembles, arr, n, C.byref(m), us), "dabgpu_multiplex_mapping")
        return m.value, {"wave": us[0], "lane": us[1], "octet": us[2]}

    def viterbi_decode_batch(self, codewords, results, tie_rule=0, stream=None):
        """codewords: list/ctypes array of Codeword (host); results: device buffer of n CodewordResult"""
        n = len(codewords)
        arr = (Codeword * n)(*codewords) if not isinstance(codewords, C.Array) else codewords
        check(lib().dabgpu_viterbi_decode_batch(self._h, arr, n, tie_rule, _ptr(results), self._stream(stream)),
              "dabgpu_viterbi_decode_batch")

    def fic_decode_frames(self, bits, n_frames, fib_bytes, results, frame_stride=NB_FRAME_BITS, tie_rule=0, stream=None):
        check(lib().dabgpu_fic_decode_frames(self._h, _ptr(bits), n_frames, frame_stride, _ptr(fib_bytes), _ptr(results),
                                             tie_rule, self._stream(stream)), "dabgpu_fic_decode_frames")

    def fic_decode_ring(self, history, n_ensembles, ensemble_stride, newest_slot, fib_bytes, results, tie_rule=0, stream=None):
        check(lib().dabgpu_fic_decode_ring(self._h, _ptr(history), n_ensembles, ensemble_stride, _ptr(newest_slot), _ptr(fib_bytes),
                                           _ptr(results), tie_rule, self._stream(stream)), "dabgpu_fic_decode_ring")

    def msc_decode_ring(self, history, n_ensembles, ensemble_stride, history_frames, newest_slot, subchannels, out, out_ensemble_stride,
                        results, tie_rule=0, stream=None, bits_layout=BITS_NATURAL):
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels)
        if bits_layout != BITS_NATURAL:
            check(lib().dabgpu_msc_decode_ring_layout(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames, _ptr(newest_slot),
                                                      arr, n, _ptr(out), out_ensemble_stride, _ptr(results), tie_rule, int(bits_layout),
                                                      self._stream(stream)), "dabgpu_msc_decode_ring_layout")
            return
        check(lib().dabgpu_msc_decode_ring(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames, _ptr(newest_slot), arr, n,
                                           _ptr(out), out_ensemble_stride, _ptr(results), tie_rule, self._stream(stream)),
              "dabgpu_msc_decode_ring")

    def decode_frames(self, history, n_ensembles, ensemble_stride, history_frames, newest_frame_slot, subchannels, fib_bytes, fic_results,
                      out, out_ensemble_stride, results, tie_rule=0, stream=None, bits_layout=BITS_NATURAL):
        """FIC (ring slot newest_frame_slot) + MSC of one transmission frame of every ensemble in one call (dabgpu_decode_frames_layout)"""
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels) if n else None
        check(lib().dabgpu_decode_frames_layout(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames, newest_frame_slot, arr, n,
                                                _ptr(fib_bytes), _ptr(fic_results), _ptr(out), out_ensemble_stride, _ptr(results), tie_rule,
                                                int(bits_layout), self._stream(stream)), "dabgpu_decode_frames_layout")

    def decode_ring(self, history, n_ensembles, ensemble_stride, history_frames, newest_slot, subchannels, fib_bytes, fic_results,
                    out, out_ensemble_stride, results, tie_rule=0, stream=None, bits_layout=BITS_NATURAL):
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels) if n else None
        check(lib().dabgpu_decode_ring_layout(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames, _ptr(newest_slot), arr, n,
                                              _ptr(fib_bytes), _ptr(fic_results), _ptr(out), out_ensemble_stride, _ptr(results), tie_rule,
                                              int(bits_layout), self._stream(stream)), "dabgpu_decode_ring_layout")

    def msc_decode_frames(self, history, n_ensembles, ensemble_stride, history_frames, newest_frame_slot, subchannels,
                          out, out_ensemble_stride, results, tie_rule=0, stream=None, bits_layout=BITS_NATURAL):
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels)
        if bits_layout != BITS_NATURAL:
            check(lib().dabgpu_msc_decode_frames_layout(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames,
                                                        newest_frame_slot, arr, n, _ptr(out), out_ensemble_stride, _ptr(results),
                                                        tie_rule, int(bits_layout), self._stream(stream)), "dabgpu_msc_decode_frames_layout")
            return
        check(lib().dabgpu_msc_decode_frames(self._h, _ptr(history), n_ensembles, ensemble_stride, history_frames,
                                             newest_frame_slot, arr, n, _ptr(out), out_ensemble_stride, _ptr(results),
                                             tie_rule, self._stream(stream)), "dabgpu_msc_decode_frames")


    def iq_convert(self, raw, fmt, n_samples, iq, stream=None):
        """device raw samples in format number `fmt` -> device interleaved float IQ (asynchronous)"""
        check(lib().dabgpu_iq_convert(self._h, _ptr(raw), int(fmt), n_samples, _ptr(iq), self._stream(stream)), "dabgpu_iq_convert")

    def iq_convert_host(self, raw, fmt):
        import numpy as np
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        sb = iq_format_sample_bytes(fmt)
        if sb == 0:
            raise DabGpuError(f"unknown IQ format number {fmt}")
        n = raw.size // sb
        out = np.empty(2 * n, np.float32)
        check(lib().dabgpu_iq_convert_host_sync(self._h, _ptr(raw), int(fmt), n, _ptr(out)), "dabgpu_iq_convert_host_sync")
        return out

    def soft_bits_to_hard_bytes(self, bits, n_bytes, out, stream=None):
        check(lib().dabgpu_soft_bits_to_hard_bytes(self._h, _ptr(bits), n_bytes, _ptr(out), self._stream(stream)),
              "dabgpu_soft_bits_to_hard_bytes")

    def hard_bytes_to_soft_bits(self, data, n_bytes, out, stream=None):
        check(lib().dabgpu_hard_bytes_to_soft_bits(self._h, _ptr(data), n_bytes, _ptr(out), self._stream(stream)),
              "dabgpu_hard_bytes_to_soft_bits")

    def soft_bits_to_hard_bytes_host(self, bits):
        import numpy as np
        bits = np.ascontiguousarray(bits, dtype=np.int8)
        out = np.empty(bits.size // 8, np.uint8)
        check(lib().dabgpu_soft_bits_to_hard_bytes_host_sync(self._h, _ptr(bits), out.size, _ptr(out)), "soft_bits_to_hard_bytes_host_sync")
        return out

    def hard_bytes_to_soft_bits_host(self, data):
        import numpy as np
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.empty(data.size * 8, np.int8)
        check(lib().dabgpu_hard_bytes_to_soft_bits_host_sync(self._h, _ptr(data), data.size, _ptr(out)), "hard_bytes_to_soft_bits_host_sync")
        return out


class StreamBank:
    """dabgpu_stream_bank: n unsynchronised receivers resident on the device (one OFDM_Demod each)"""

    def __init__(self, ctx, n_streams, cfg=None, mode=1):
        self._ctx = ctx
        self.n = n_streams
        self.mode = mode
        self._h = C.c_void_p()
        check(lib().dabgpu_stream_bank_create_mode(ctx._h, int(mode), n_streams, C.byref(cfg) if cfg is not None else None, C.byref(self._h)),
              "dabgpu_stream_bank_create_mode")

    def close(self):
        if self._h:
            lib().dabgpu_stream_bank_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=None):
        check(lib().dabgpu_stream_bank_reset(self._h, Context._stream(stream)), "dabgpu_stream_bank_reset")

    def process(self, iq, stream_stride_samples, n_samples, bits, max_frames, n_frames=None, stream=None):
        check(lib().dabgpu_stream_bank_process(self._h, _ptr(iq), stream_stride_samples, n_samples, _ptr(bits), max_frames,
                                               _ptr(n_frames), Context._stream(stream)), "dabgpu_stream_bank_process")

    def process_raw(self, raw, fmt, stream_stride_samples, n_samples, bits, max_frames, n_frames=None, stream=None):
        check(lib().dabgpu_stream_bank_process_raw(self._h, _ptr(raw), int(fmt), stream_stride_samples, n_samples, _ptr(bits), max_frames,
                                                   _ptr(n_frames), Context._stream(stream)), "dabgpu_stream_bank_process_raw")

    def process_retained(self, raw, fmt, stream_stride_samples, n_samples, prev_raw, bits, max_frames, n_frames=None, stream=None):
        """process_raw for callers that keep `raw` valid and unchanged until the NEXT call has returned: no carry-over copy at the end of
        the block; prev_raw = the block of the previous retained call (None at the first)"""
        check(lib().dabgpu_stream_bank_process_retained(self._h, _ptr(raw), int(fmt), stream_stride_samples, n_samples, _ptr(prev_raw), _ptr(bits),
                                                        max_frames, _ptr(n_frames), Context._stream(stream)), "dabgpu_stream_bank_process_retained")

    def release(self, prev_raw, fmt, stream_stride_samples, stream=None):
        check(lib().dabgpu_stream_bank_release(self._h, _ptr(prev_raw), int(fmt), stream_stride_samples, Context._stream(stream)),
              "dabgpu_stream_bank_release")

    def process_ring(self, raw, fmt, stream_stride_samples, n_samples, hist, hist_frames, newest_slot, stream=None, bits_layout=BITS_NATURAL):
        if bits_layout != BITS_NATURAL:
            check(lib().dabgpu_stream_bank_process_ring_layout(self._h, _ptr(raw), int(fmt), stream_stride_samples, n_samples, _ptr(hist),
                                                               hist_frames, _ptr(newest_slot), int(bits_layout), Context._stream(stream)),
                  "dabgpu_stream_bank_process_ring_layout")
            return
        check(lib().dabgpu_stream_bank_process_ring(self._h, _ptr(raw), int(fmt), stream_stride_samples, n_samples, _ptr(hist), hist_frames,
                                                    _ptr(newest_slot), Context._stream(stream)), "dabgpu_stream_bank_process_ring")

    def process_ring_retained(self, raw, fmt, stream_stride_samples, n_samples, prev_raw, hist, hist_frames, newest_slot, stream=None, bits_layout=BITS_NATURAL):
        """process_ring for callers that keep `raw` valid and unchanged until the NEXT call has returned (no carry-over copy);
        prev_raw = the block of the previous retained call (None at the first)"""
        check(lib().dabgpu_stream_bank_process_ring_retained(self._h, _ptr(raw), int(fmt), stream_stride_samples, n_samples, _ptr(prev_raw), _ptr(hist),
                                                             hist_frames, _ptr(newest_slot), int(bits_layout), Context._stream(stream)),
              "dabgpu_stream_bank_process_ring_retained")

    def status(self, stream=None):
        import numpy as np
        out = np.zeros(self.n, dtype=np.dtype(STREAM_STATUS_DTYPE))
        check(lib().dabgpu_stream_bank_status(self._h, _ptr(out), Context._stream(stream)), "dabgpu_stream_bank_status")
        return out


class IngestPipe:
    """dabgpu_ingest: ring of pinned host buffers + device twins + a copy stream (include/dabgpu.h)"""

    def __init__(self, ctx, buffer_bytes, depth=2):
        self._ctx = ctx
        self.bytes = buffer_bytes
        self._h = C.c_void_p()
        check(lib().dabgpu_ingest_create(ctx._h, buffer_bytes, depth, C.byref(self._h)), "dabgpu_ingest_create")

    def close(self):
        if self._h:
            lib().dabgpu_ingest_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def acquire(self):
        """-> numpy uint8 view of the next pinned buffer"""
        import numpy as np
        p = C.c_void_p()
        check(lib().dabgpu_ingest_acquire(self._h, C.byref(p)), "dabgpu_ingest_acquire")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.bytes,))

    def submit(self, n_bytes):
        """starts the copy of the acquired buffer -> device pointer (int) of its twin"""
        d = C.c_void_p()
        check(lib().dabgpu_ingest_submit(self._h, n_bytes, C.byref(d)), "dabgpu_ingest_submit")
        return d.value

    def wait(self, d_buffer, stream=None):
        check(lib().dabgpu_ingest_wait(self._h, C.c_void_p(d_buffer), Context._stream(stream)), "dabgpu_ingest_wait")

    def consumed(self, d_buffer, stream=None):
        check(lib().dabgpu_ingest_consumed(self._h, C.c_void_p(d_buffer), Context._stream(stream)), "dabgpu_ingest_consumed")


class FrameSession:
    """dabgpu_frame_session: one receiver's frames decoded one batched device call each (FIC + every registered sub-channel), results
    fetched by (generation, FIB group / sub-channel, CIF) -- what the mirror classes use through dabgpu_frame_batcher"""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        L = lib()
        L.dabgpu_frame_session_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.dabgpu_frame_session_destroy.argtypes = [C.c_void_p]
        L.dabgpu_frame_session_destroy.restype = None
        L.dabgpu_frame_session_set_subchannels.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.dabgpu_frame_session_push_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.dabgpu_frame_session_fetch_fib_group.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.dabgpu_frame_session_fetch_cif.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                                     C.POINTER(C.c_uint64)]
        check(L.dabgpu_frame_session_create(C.byref(self._h), device), "dabgpu_frame_session_create")

    def close(self):
        if self._h:
            lib().dabgpu_frame_session_destroy(self._h)
            self._h = C.c_void_p()

    def set_subchannels(self, subchannels):
        n = len(subchannels)
        arr = (SubChannel * n)(*subchannels) if n else None
        check(lib().dabgpu_frame_session_set_subchannels(self._h, arr, n), "dabgpu_frame_session_set_subchannels")

    def push_frame(self, bits, decode_fic=True, tie_rule=0):
        """bits: numpy int8[230400] (host) -> generation number of the frame"""
        import numpy as np
        b = np.ascontiguousarray(bits, dtype=np.int8)
        assert b.size == NB_FRAME_BITS
        gen = C.c_uint64()
        check(lib().dabgpu_frame_session_push_frame(self._h, _ptr(b), int(bool(decode_fic)), tie_rule, C.byref(gen)), "dabgpu_frame_session_push_frame")
        return gen.value

    def fetch_fib_group(self, generation, group):
        """-> (bytes uint8[96], crc_ok_mask, path_error), or None when that generation is gone / was pushed without the FIC"""
        import numpy as np
        out = np.empty(96, dtype=np.uint8)
        m, e = C.c_uint32(), C.c_uint64()
        st = lib().dabgpu_frame_session_fetch_fib_group(self._h, generation, group, _ptr(out), C.byref(m), C.byref(e))
        if st == 4:                                   # DABGPU_ERR_NOT_READY
            return None
        check(st, "dabgpu_frame_session_fetch_fib_group")
        return out, m.value, e.value

    def fetch_cif(self, generation, subchannel, cif, capacity=8192):
        """-> (bytes uint8[n], path_error), or None (generation gone / sub-channel not registered when that frame was pushed)"""
        import numpy as np
        out = np.empty(capacity, dtype=np.uint8)
        n, e = C.c_size_t(), C.c_uint64()
        st = lib().dabgpu_frame_session_fetch_cif(self._h, generation, C.byref(subchannel), cif, _ptr(out), capacity, C.byref(n), C.byref(e))
        if st == 4:
            return None
        check(st, "dabgpu_frame_session_fetch_cif")
        return out[:n.value].copy(), e.value


class DabPlusBank:
    """dabgpu_dabplus_bank: n AAC_Frame_Processor states resident on the device"""

    def __init__(self, ctx, n_streams):
        self._ctx = ctx
        self.n = n_streams
        self._h = C.c_void_p()
        check(lib().dabgpu_dabplus_bank_create(ctx._h, n_streams, C.byref(self._h)), "dabgpu_dabplus_bank_create")

    def close(self):
        if self._h:
            lib().dabgpu_dabplus_bank_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=None):
        check(lib().dabgpu_dabplus_bank_reset(self._h, Context._stream(stream)), "dabgpu_dabplus_bank_reset")

    def process(self, frames, stream_offsets, frame_stride, frame_bytes, n_frames, superframes, superframe_stride, results,
                max_superframes, counts, stream=None):
        check(lib().dabgpu_dabplus_bank_process(self._h, _ptr(frames), _ptr(stream_offsets), frame_stride, _ptr(frame_bytes),
                                                n_frames, _ptr(superframes), superframe_stride, _ptr(results), max_superframes,
                                                _ptr(counts), Context._stream(stream)), "dabgpu_dabplus_bank_process")

    def process_masked(self, frames, stream_offsets, frame_stride, frame_bytes, n_frames, superframes, superframe_stride, results,
                       max_superframes, counts, active, streams_per_flag, stream=None):
        check(lib().dabgpu_dabplus_bank_process_masked(self._h, _ptr(frames), _ptr(stream_offsets), frame_stride, _ptr(frame_bytes),
                                                       n_frames, _ptr(superframes), superframe_stride, _ptr(results), max_superframes,
                                                       _ptr(counts), _ptr(active), streams_per_flag, Context._stream(stream)),
              "dabgpu_dabplus_bank_process_masked")

    def process_frame_host(self, frame):
        """one-stream bank: -> (superframe_done, firecode_wait_failed, result record, super frame bytes or None)"""
        import numpy as np
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        done, wait = C.c_int(0), C.c_int(0)
        res = np.zeros(1, dtype=np.dtype(SUPERFRAME_RESULT_DTYPE))
        res["rs_failed_index"] = -1                      # the record is only written when a super frame was attempted
        res["au_walk_stopped_at"] = -1
        sf = np.zeros(5 * frame.size, np.uint8)
        check(lib().dabgpu_dabplus_process_frame_host_sync(self._h, _ptr(frame), frame.size, C.byref(done), C.byref(wait), None, _ptr(res),
                                                           _ptr(sf)), "dabgpu_dabplus_process_frame_host_sync")
        return done.value, wait.value, res[0], (sf if done.value else None)


def stream_cfg_default():
    c = StreamCfg()
    lib().dabgpu_stream_cfg_default(C.byref(c))
    return c


def ofdm_params(mode):
    """dict of the geometry of transmission mode 1..4 (host only)"""
    out = (C.c_int * 9)()
    check(lib().dabgpu_get_ofdm_params(int(mode), out), "dabgpu_get_ofdm_params")
    keys = ("nb_frame_symbols", "nb_symbol_period", "nb_null_period", "nb_fft", "nb_cyclic_prefix", "nb_data_carriers",
            "nb_frame_samples", "nb_sym_bits", "nb_frame_bits")
    return dict(zip(keys, list(out)))


def carrier_mapper(mode):
    import numpy as np
    m = np.zeros(ofdm_params(mode)["nb_data_carriers"], np.int32)
    check(lib().dabgpu_get_carrier_mapper(int(mode), _ptr(m)), "dabgpu_get_carrier_mapper")
    return m


def iq_format_from_mode(mode):
    return lib().dabgpu_iq_format_from_mode(mode.encode())


def iq_format_sample_bytes(fmt):
    return int(lib().dabgpu_iq_format_sample_bytes(int(fmt)))


def wav_parse_header(image):
    """host-only: WavHeader of a file image (numpy uint8 / bytes); raises DabGpuError where the reference's reader throws"""
    import numpy as np
    image = np.frombuffer(bytes(image), np.uint8) if isinstance(image, (bytes, bytearray)) else np.ascontiguousarray(image, np.uint8)
    h = WavHeader()
    check(lib().dabgpu_wav_parse_header(_ptr(image), image.size, C.byref(h)), "dabgpu_wav_parse_header")
    return h


def sync_cfg_default():
    c = SyncCfg()
    lib().dabgpu_sync_cfg_default(C.byref(c))
    return c


def subchannel_plan(sc):
    """(pi[], l[], n_decoded_bytes) of a SubChannel via the product's host tables"""
    pi = (C.c_int * 4)()
    lx = (C.c_int * 4)()
    nb = C.c_int(0)
    n = lib().dabgpu_subchannel_plan(C.byref(sc), pi, lx, C.byref(nb))
    if n < 0:
        raise DabGpuError("invalid sub-channel protection profile")
    return list(pi)[:n], list(lx)[:n], nb.value
