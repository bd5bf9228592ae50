"""ctypes binding of libtwstft_hip.so (C ABI: include/twstft_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module
raises, and every product entry point that needs it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWX_LIB") or os.path.join(_HERE, "libtwstft_hip.so")   # TWX_LIB: kernel-variant experiments

TWX_OK = 0
TWX_E_SIZE = -2
TWX_CONV_GODUAL, TWX_CONV_CLAUDIO = 0, 1
TWX_WIN_NONE, TWX_WIN_HAMMING = 0, 1
TWX_F32, TWX_F64 = 0, 1
TWX_OPT_REMOVE_MEAN = 1
TWX_OPT_FIR_MFMA = 2
TWX_OPT_SELFCHECK = 3
TWX_OPT_BRUIT_LEN, TWX_OPT_NOISE_SQUARE_LEN = 4, 5
TWX_OPT_DEBUG_FAULT = 102
TWX_STATUS_SELFCHECK = 1
TWX_STATUS_RESAMPLE_NAN = 2
TWX_FLAG_PROFILE = 1
TWX_FLAG_FINE_FREQ = 2
TWX_FLAG_CODE_ZERO_MEAN = 4
TWX_CODE_BIPOLAR, TWX_CODE_UNIPOLAR = 0, 1
TWX_PROF_MAX = 16
TWX_TRK_RANGING, TWX_TRK_RE, TWX_TRK_LO = 0, 1, 2
TWX_CARRIER_SEARCH_DF, TWX_CARRIER_CHUNK_BAND = 0, 1
TWX_ABI_VERSION = 6
TWX_MULTI_NO_RCCL, TWX_MULTI_RCCL_ONE = 1, 2
TWX_ACQ_IZAMAX = 1


def TWX_ACQ_DEC(d: int) -> int:
    return (int(d) & 0xFF) << 8


class TwxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"twstft_hip error {status}: {msg}")
        self.status = status


class twx_config(C.Structure):
    _fields_ = [("fs", C.c_double), ("sps", C.c_int32), ("nint", C.c_int32),
                ("chips", C.POINTER(C.c_uint8)), ("n_chips", C.c_int64),
                ("lfsr_bitlen", C.c_int32), ("lfsr_taps", C.c_int32),
                ("convention", C.c_int32), ("window", C.c_int32), ("precision", C.c_int32),
                ("var_ddof", C.c_int32), ("snr_rot", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("flags", C.c_int32), ("chips_q", C.POINTER(C.c_uint8)),
                ("code_levels", C.c_int32), ("nphase", C.c_int32)]


class twx_band(C.Structure):
    _fields_ = [("k_lo", C.c_int64), ("k_hi", C.c_int64)]


class twx_result(C.Structure):
    _fields_ = [("indice0", C.c_int64), ("correction", C.c_double),
                ("xval", C.c_double * 2), ("xvalm1", C.c_double * 2), ("xvalp1", C.c_double * 2),
                ("zwin", (C.c_double * 2) * 7),
                ("df", C.c_double), ("df_index", C.c_int64),
                ("SNRr", C.c_double), ("SNRi", C.c_double), ("puissance", C.c_double),
                ("puissancecode", C.c_double), ("puissancenoise", C.c_double),
                ("status", C.c_int32), ("dt", C.c_int32)]


class twx_extra(C.Structure):
    _fields_ = [("bruit", C.c_double), ("valmax_square", C.c_double), ("noise_square", C.c_double), ("reserved", C.c_double)]


class twx_info(C.Structure):
    _fields_ = [("n", C.c_int64), ("n1", C.c_int32), ("n2", C.c_int32), ("nphase", C.c_int32),
                ("batch", C.c_int32), ("precision", C.c_int32), ("col_w", C.c_int32),
                ("device_bytes", C.c_int64)]


class twx_tracked_config(C.Structure):
    _fields_ = [("fs", C.c_double), ("sps", C.c_int32), ("nint", C.c_int32),
                ("chips", C.POINTER(C.c_uint8)), ("n_chips", C.c_int64),
                ("lfsr_bitlen", C.c_int32), ("lfsr_taps", C.c_int32), ("chunk_samples", C.c_int64),
                ("band_lo_hz", C.c_double), ("band_hi_hz", C.c_double), ("carrier", C.c_int32), ("indice_floor", C.c_int32),
                ("df_threshold", C.c_double), ("skip_samples", C.c_int64), ("precision", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("reserved", C.c_int32)]


class twx_tracked_code(C.Structure):
    _fields_ = [("xval", C.c_double * 2), ("indice1", C.c_double), ("correction1", C.c_double), ("SNR1r", C.c_double),
                ("SNR1i", C.c_double), ("puissance1", C.c_double)]


class twx_tracked_summary(C.Structure):
    _fields_ = [("n_codes", C.c_int64), ("n_chunks", C.c_int64), ("n_moved", C.c_int64), ("kbon", C.c_int64),
                ("batches", C.c_int64), ("puissancecode", C.c_double), ("puissancenoise", C.c_double)]


class twx_acq_result(C.Structure):
    _fields_ = [("fc", C.c_double), ("pk", C.c_double), ("pt", C.c_int64), ("n_trials", C.c_int64)]


class twx_track_state(C.Structure):
    _fields_ = [("fs", C.c_double), ("duration", C.c_double), ("psbb", C.c_double), ("fc", C.c_double), ("df", C.c_double),
                ("phi", C.c_double), ("last_phi", C.c_double), ("pt", C.c_int64), ("fc_prev", C.c_double), ("pt_prev", C.c_int64)]


class twx_track_result(C.Structure):
    _fields_ = [("freq", C.c_double), ("phi", C.c_double), ("gd", C.c_double), ("dg", C.c_double), ("sdgd", C.c_double),
                ("pk", C.c_double), ("cnt", C.c_int32), ("updated", C.c_int32)]


class twx_multi_info(C.Structure):
    _fields_ = [("n_contexts", C.c_int32), ("n_devices_distinct", C.c_int32), ("rccl", C.c_int32), ("rccl_version", C.c_int32),
                ("records_gathered", C.c_int64), ("bytes_per_rank", C.c_int64), ("gather_ms", C.c_double),
                ("rccl_fallback", C.c_int32), ("threads_pinned", C.c_int32), ("numa_node", C.c_int32 * 64), ("rccl_error", C.c_char * 256)]


class twx_track_mai(C.Structure):
    _fields_ = [("pk_idx", C.POINTER(C.c_int32)), ("amp", C.POINTER(C.c_double)), ("phase", C.POINTER(C.c_double))]


class twx_rx_row(C.Structure):
    _fields_ = [("ch", C.c_char), ("mode", C.c_char), ("reserved", C.c_int16), ("pn", C.c_int32), ("fc_init", C.c_double), ("kcps", C.c_int32),
                ("reserved2", C.c_int32), ("fltkhz", C.c_double), ("frange", C.c_double), ("fstep", C.c_double), ("snr_min_db", C.c_double),
                ("code", C.POINTER(C.c_uint8)), ("code_len", C.c_int64)]


class twx_rx_config(C.Structure):
    _fields_ = [("fs_in", C.c_double), ("ninterp", C.c_int32), ("dec_a", C.c_int32), ("code_dir", C.c_char_p), ("out_dir", C.c_char_p),
                ("seed", C.c_uint64), ("acq_block", C.c_int32), ("device", C.c_int32)]


class twx_rx_report(C.Structure):
    _fields_ = [("status", C.c_int32), ("cnt", C.c_int32), ("fc", C.c_double), ("df", C.c_double), ("phi", C.c_double), ("gd", C.c_double),
                ("dg", C.c_double), ("sdgd", C.c_double), ("pk", C.c_double), ("px", C.c_double), ("pt", C.c_int64), ("acq_idx", C.c_int64),
                ("n_trials", C.c_int64), ("dat_row", C.c_char * 128)]


class twx_rx_channel_info(C.Structure):
    _fields_ = [("pn", C.c_int32), ("is_chA", C.c_int32), ("clen", C.c_int32), ("nlag", C.c_int32), ("bps", C.c_int32), ("is_sic", C.c_int32),
                ("nobs", C.c_int64), ("nfft", C.c_int64), ("duration", C.c_double), ("range", C.c_double), ("step", C.c_double),
                ("snr_min", C.c_double), ("psbb", C.c_double), ("dat_name", C.c_char * 64)]


TWX_RX_NO_SIGNAL, TWX_RX_ACQUIRED, TWX_RX_CODE_LOCK, TWX_RX_TRACKED, TWX_RX_ACQ_FAILED, TWX_RX_LOCK_LOST = range(6)


class twx_prof_entry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms_total", C.c_double), ("launches", C.c_int64), ("units", C.c_int64)]


# every symbol include/twstft_hip.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "twx_abi_version": (C.c_int, []),
    "twx_strerror": (C.c_char_p, [C.c_int]),
    "twx_last_error": (C.c_char_p, [_VP]),
    "twx_plan_source_hash": (C.c_char_p, []),
    "twx_load_plan": (C.c_int, [C.c_char_p]),
    "twx_plan_available": (C.c_int, [C.c_int64, C.c_int32]),
    "twx_plan_lengths": (C.c_int, [C.c_int32, C.c_int32, _VP, _VP, C.c_int32]),
    "twx_create": (C.c_int, [C.POINTER(twx_config), C.POINTER(_VP)]),
    "twx_destroy": (None, [_VP]),
    "twx_get_info": (C.c_int, [_VP, C.POINTER(twx_info)]),
    "twx_process_windows": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(twx_band), _VP, _VP]),
    "twx_process_windows_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(twx_band), _VP, _VP]),
    "twx_process_complex": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, C.POINTER(twx_band), _VP, _VP]),
    "twx_synchronize": (C.c_int, [_VP]),
    "twx_set_option": (C.c_int, [_VP, C.c_int32, C.c_int64]),
    "twx_stream": (_VP, [_VP]),
    "twx_fetch_extra": (C.c_int, [_VP, C.POINTER(twx_extra), C.c_int64]),
    "twx_set_resample": (C.c_int, [_VP, C.c_double, C.c_double, C.c_int64]),
    "twx_get_resample": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "twx_selfcheck_stats": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "twx_fft_forward": (C.c_int, [_VP, _VP, _VP]),
    "twx_get_code_spectrum": (C.c_int, [_VP, _VP]),
    "twx_xcorr_map": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_double, _VP]),
    "twx_process_file": (C.c_int, [_VP, C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(twx_band), C.c_double, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "twx_set_code_spectrum": (C.c_int, [_VP, _VP]),
    "twx_xcorr_map_dev": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_double, _VP]),
    "twx_caf_freqs_cdev": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int32, _VP]),
    "twx_acquire_cdev": (C.c_int, [_VP, _VP, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.POINTER(twx_acq_result)]),
    "twx_caf_bins": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _VP, _VP]),
    "twx_caf_bins_dev": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _VP, _VP]),
    "twx_caf_freqs": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, _VP, C.c_int64, _VP]),
    "twx_sqspec_bins_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_int32, _VP]),
    "twx_sqspec_band_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, _VP]),
    "twx_sliding_dot": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double, C.c_double, C.c_double, _VP]),
    "twx_track_update": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.POINTER(twx_track_state), C.POINTER(twx_track_result)]),
    "twx_track_epoch_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double,
                                      C.POINTER(twx_track_state), C.POINTER(twx_track_result)]),
    "twx_fir_decimate": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_int32, C.c_int32, _VP, _VP, C.POINTER(C.c_int64)]),
    "twx_sliding_dot_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double, C.c_double, C.c_double, _VP]),
    "twx_fir_decimate_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_int32, C.c_int32, _VP, _VP, C.POINTER(C.c_int64)]),
    "twx_tracked_defaults": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.POINTER(twx_tracked_config)]),
    "twx_tracked_create": (C.c_int, [C.POINTER(twx_tracked_config), C.POINTER(_VP)]),
    "twx_tracked_destroy": (None, [_VP]),
    "twx_tracked_last_error": (C.c_char_p, [_VP]),
    "twx_tracked_context": (_VP, [_VP]),
    "twx_tracked_file": (C.c_int, [_VP, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(twx_tracked_summary)]),
    "twx_tracked_host": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.POINTER(twx_tracked_summary)]),
    "twx_tracked_fetch": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "twx_tracked_timing": (C.c_int, [_VP, _VP, _VP]),
    "twx_tracked_search_df": (C.c_int, [_VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "twx_multi_create": (C.c_int, [C.POINTER(twx_config), _VP, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "twx_multi_destroy": (None, [_VP]),
    "twx_multi_last_error": (C.c_char_p, [_VP]),
    "twx_multi_get_info": (C.c_int, [_VP, C.POINTER(twx_multi_info)]),
    "twx_multi_context": (_VP, [_VP, C.c_int32]),
    "twx_multi_process_file": (C.c_int, [_VP, C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(twx_band), C.c_double, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "twx_multi_process_windows": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(twx_band), _VP, _VP]),
    "twx_multi_process_windows_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(twx_band), _VP, _VP]),
    "twx_multi_fetch_gathered": (C.c_int, [_VP, C.c_int32, _VP, C.c_int64]),
    "twx_multi_block": (C.c_int, [_VP, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "twx_multi_process_recording_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(twx_band), _VP, _VP]),
    "twx_multi_exchange_only": (C.c_int, [_VP, C.c_int64]),
    "twx_file_df": (C.c_int, [C.c_char_p, C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "twx_write_cmat": (C.c_int, [C.c_char_p, _VP, _VP, C.c_int64]),
    "twx_file_df_last_error": (C.c_char_p, []),
    "twx_device_affinity": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]),
    "twx_pin_thread_to_device": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "twx_set_code_spectrum_dev": (C.c_int, [_VP, _VP]),
    "twx_fft_forward_dev": (C.c_int, [_VP, _VP, _VP]),
    "twx_sliding_dot_cdev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double, C.c_double, C.c_double, _VP]),
    "twx_track_epoch_cdev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double, C.POINTER(twx_track_state), C.POINTER(twx_track_result)]),
    "twx_track_epoch_cdev_mai": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _VP, C.c_double, C.POINTER(twx_track_state), C.POINTER(twx_track_result),
                                          C.POINTER(twx_track_mai)]),
    "twx_track_update_mai": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int64, C.POINTER(twx_track_state), C.POINTER(twx_track_result), C.POINTER(twx_track_mai)]),
    "twx_rx_parse_param": (C.c_int, [C.c_char_p, C.POINTER(twx_rx_row), C.c_int32]),
    "twx_rx_create": (C.c_int, [C.POINTER(twx_rx_config), C.POINTER(twx_rx_row), C.c_int32, C.POINTER(_VP)]),
    "twx_rx_destroy": (None, [_VP]),
    "twx_rx_last_error": (C.c_char_p, [_VP]),
    "twx_rx_channel": (C.c_int, [_VP, C.c_int32, C.POINTER(twx_rx_channel_info)]),
    "twx_rx_second": (C.c_int, [_VP, _VP, C.POINTER(twx_rx_report)]),
    "twx_rx_second_dev": (C.c_int, [_VP, _VP, C.POINTER(twx_rx_report)]),
    "twx_rx_file": (C.c_int, [_VP, C.c_char_p, C.c_int64, C.POINTER(twx_rx_report), C.c_int64, C.POINTER(C.c_int64)]),
    "twx_rx_stream_dev": (_VP, [_VP, C.c_int32]),
    "twx_rx_powers": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "twx_rx_console_line": (C.c_int, [_VP, C.c_int32, C.POINTER(twx_rx_report), C.c_char_p, C.c_int32]),
    "twx_debug_stamps": (C.c_int, [_VP, _VP, C.c_longlong]),
    "twx_profile_reset": (C.c_int, [_VP]),
    "twx_profile_get": (C.c_int, [_VP, C.POINTER(twx_prof_entry), C.c_int32, C.POINTER(C.c_int32)]),
    "twx_lfsr_chips": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _VP]),
    "twx_synth_capture_dev": (C.c_int, [_VP, C.c_int64, C.c_int64, _VP, C.c_int64, C.c_int32, C.c_int32, _VP, _VP]),
    "twx_ctx_alloc": (_VP, [_VP, C.c_size_t]),
    "twx_ctx_free": (None, [_VP, _VP]),
    "twx_dev_alloc": (_VP, [C.c_size_t]),
    "twx_dev_free": (None, [_VP]),
    "twx_memcpy_h2d": (C.c_int, [_VP, _VP, C.c_size_t]),
    "twx_memcpy_d2h": (C.c_int, [_VP, _VP, C.c_size_t]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64.so.1.  If this library pulled in /opt/rocm's copies first, a later
    # `import torch` would find "No HIP GPUs".  Loading torch's copies first (when torch is
    # installed) makes both users share them; hosts without torch (MATLAB/Octave MEX) are unaffected.
    if os.environ.get("TWX_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.twx_abi_version() != TWX_ABI_VERSION:
        raise ImportError("libtwstft_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status, ctx=None):
    if status != TWX_OK:
        lib = load()
        msg = lib.twx_last_error(ctx)
        raise TwxError(status, (msg or lib.twx_strerror(status) or b"?").decode())
