"""Several GPUs from ONE host process: binding of ``twx_multi_*`` (include/twstft_hip.h).

A MATLAB / Octave / C host is one process; its route to more than one GPU is the library's own driver — one correlator
context and one host thread per device, contiguous blocks of windows (the sharding rule of :mod:`dist`), one
``ncclAllGather`` of the fixed-size records over xGMI (RCCL bound by the library itself, no ``torch.distributed``) — the
reference's analogue being the worker threads of processing/CPP/main.cpp:180-187,488-497 and the side-by-side jobs of
acquisition/goprocess.sh:9-11.  A device list that repeats a device (``[0, 0, 0, 0]``) runs the same threads and the same
ordering with the blocks concatenated on the host, which is how a one-GPU box tests it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .correlator import _to_result


class MultiCorrelator:
    """``Correlator`` over a list of devices.  ``devices``: ordinals (repeats allowed) or an int N = devices 0..N-1 modulo the
    visible ones.  ``rccl``: ``None`` = RCCL when the devices are distinct and more than one, ``False`` = never (host-side
    concatenation), ``True`` = also for a single device (a world of one, same calls)."""

    def __init__(self, chips=None, devices=1, fs: float = 5e6, sps: int = 2, Nint: int = 1, *, lfsr=None, var_ddof: int = 0,
                 snr_rot: int = -1, convention: str = "godual", precision: str = "f32", max_batch: int = 0, rccl: bool | None = None):
        self._lib = L.load()
        cfg = L.twx_config()
        cfg.fs, cfg.sps, cfg.nint = fs, sps, Nint
        if chips is not None:
            self._chips = np.ascontiguousarray(chips, dtype=np.uint8)
            cfg.chips = self._chips.ctypes.data_as(C.POINTER(C.c_uint8))
            cfg.n_chips = self._chips.size
        elif lfsr is not None:
            cfg.lfsr_bitlen, cfg.lfsr_taps, cfg.n_chips = lfsr
        else:
            raise ValueError("give chips or lfsr=(bitlen, taps, noiselen)")
        cfg.convention = {"godual": L.TWX_CONV_GODUAL, "claudio": L.TWX_CONV_CLAUDIO}[convention]
        cfg.precision = {"f32": L.TWX_F32, "f64": L.TWX_F64}[precision]
        cfg.var_ddof, cfg.snr_rot, cfg.max_batch, cfg.device = var_ddof, snr_rot, max_batch, -1
        if isinstance(devices, int):
            n, dptr = int(devices), None
        else:
            self._devs = np.ascontiguousarray(devices, dtype=np.int32)
            n, dptr = self._devs.size, self._devs.ctypes.data_as(C.c_void_p)
        flags = 0 if rccl is None else (L.TWX_MULTI_RCCL_ONE if rccl else L.TWX_MULTI_NO_RCCL)
        h = C.c_void_p()
        rc = self._lib.twx_multi_create(C.byref(cfg), dptr, n, flags, C.byref(h))
        if rc == L.TWX_E_SIZE:
            from . import plans
            plans.ensure(int(cfg.n_chips) * int(sps), int(cfg.precision), self._lib)
            rc = self._lib.twx_multi_create(C.byref(cfg), dptr, n, flags, C.byref(h))
        if rc:
            raise L.TwxError(rc, (self._lib.twx_multi_last_error(None) or b"?").decode())
        self._h = h
        self.n_contexts = n
        info = L.twx_info()
        L.check(self._lib.twx_get_info(self._lib.twx_multi_context(self._h, 0), C.byref(info)))
        self.n = int(info.n)
        self.fs, self.Nint = fs, Nint

    def _check(self, rc):
        if rc:
            raise L.TwxError(rc, (self._lib.twx_multi_last_error(self._h) or b"?").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.twx_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self) -> L.twx_multi_info:
        i = L.twx_multi_info()
        self._check(self._lib.twx_multi_get_info(self._h, C.byref(i)))
        return i

    @staticmethod
    def _band(band):
        if band is None:
            return None
        b = L.twx_band(int(band[0]), int(band[1]))
        return C.byref(b)

    def process_file(self, path: str, n_channels=1, channel=0, band=None, df=None, skip_samples: int = 0,
                     max_windows: int | None = None, raw_records: bool = False):
        """``Correlator.process_file`` over all devices (same records, same order)."""
        per = self.n * 4 * n_channels
        try:
            avail = max(0, (os.path.getsize(path) - skip_samples * 4 * n_channels)) // per
        except OSError:
            avail = 0 if max_windows is None else max_windows            # the library reports the missing file
        nmax = avail if max_windows is None else min(avail, max_windows)
        nch_out = n_channels if channel < 0 else 1
        out = (L.twx_result * max(nmax * nch_out, 1))()
        if band is None and df is None:
            raise ValueError("give band (estimate df) or df")
        ndone = C.c_int64()
        self._check(self._lib.twx_multi_process_file(self._h, os.fsencode(path), n_channels, channel, skip_samples, self._band(band),
                                                     float(df) if df is not None else 0.0, C.cast(out, C.c_void_p), nmax, C.byref(ndone)))
        if raw_records:
            return np.frombuffer(bytes(out), dtype=np.uint8).reshape(-1, C.sizeof(L.twx_result))[:ndone.value * nch_out].copy()
        if channel < 0:
            return {c: [_to_result(out[w * n_channels + c]) for w in range(ndone.value)] for c in range(n_channels)}
        return [_to_result(out[i]) for i in range(ndone.value)]

    def process(self, raw, n_channels=1, channel=0, band=None, df=None, raw_records: bool = False):
        """``Correlator.process`` (capture in host memory) over all devices."""
        raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1)
        nwin = raw.size // (self.n * 2 * n_channels)
        nch_out = n_channels if channel < 0 else 1
        out = (L.twx_result * max(nwin * nch_out, 1))()
        dptr = None
        if band is None:
            if df is None:
                raise ValueError("give band (estimate df) or df (per window)")
            dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), (nwin, n_channels) if channel < 0 else (nwin,)))
            dptr = dfa.ctypes.data_as(C.c_void_p)
        self._check(self._lib.twx_multi_process_windows(self._h, raw.ctypes.data_as(C.c_void_p), nwin, n_channels, channel,
                                                        self._band(band), dptr, C.cast(out, C.c_void_p)))
        if raw_records:
            return np.frombuffer(bytes(out), dtype=np.uint8).reshape(-1, C.sizeof(L.twx_result))[:nwin * nch_out].copy()
        if channel < 0:
            return {c: [_to_result(out[w * n_channels + c]) for w in range(nwin)] for c in range(n_channels)}
        return [_to_result(out[i]) for i in range(nwin)]

    def process_dev(self, iq_dev_ptrs, nwin: int, n_channels=1, channel=0, band=None, df=None, fetch: bool = True):
        """Context i processes ``nwin`` windows of its own device-resident recording ``iq_dev_ptrs[i]``; returns the gathered
        records as a uint8 array [n_contexts*nwin*(channels), sizeof(twx_result)] (``fetch=False``: nothing, timing loops)."""
        ptrs = (C.c_void_p * self.n_contexts)(*[int(p) for p in iq_dev_ptrs])
        per = n_channels if channel < 0 else 1
        nrec = self.n_contexts * nwin * per
        dptr = None
        if band is None:
            dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), (nwin * per,)))
            dptr = dfa.ctypes.data_as(C.c_void_p)
        out = np.empty((nrec, C.sizeof(L.twx_result)), dtype=np.uint8) if fetch else None
        self._check(self._lib.twx_multi_process_windows_dev(self._h, C.cast(ptrs, C.c_void_p), nwin, n_channels, channel, self._band(band), dptr,
                                                            out.ctypes.data_as(C.c_void_p) if fetch else None))
        return out

    def block(self, n_windows_total: int, i: int) -> tuple[int, int]:
        """(start, count) of context ``i``'s contiguous block of a recording of ``n_windows_total`` windows (``dist.shard_windows``)."""
        s, c = C.c_int64(), C.c_int64()
        self._check(self._lib.twx_multi_block(self._h, int(n_windows_total), int(i), C.byref(s), C.byref(c)))
        return int(s.value), int(c.value)

    def process_recording_dev(self, iq_block_ptrs, n_windows_total: int, n_channels=1, channel=0, band=None, df=None, fetch: bool = True):
        """BASELINE.json configs[3] as written: ONE recording of ``n_windows_total`` windows, context i holding its block
        (:meth:`block`) at ``iq_block_ptrs[i]`` on its device; one exchange; returns the records in window order as a uint8
        array [n_windows_total*(channels), sizeof(twx_result)] (``fetch=False``: nothing, timing loops)."""
        ptrs = (C.c_void_p * self.n_contexts)(*[int(p) if p else None for p in iq_block_ptrs])
        per = n_channels if channel < 0 else 1
        dptr = None
        if band is None:
            dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), (n_windows_total * per,)))
            dptr = dfa.ctypes.data_as(C.c_void_p)
        out = np.empty((n_windows_total * per, C.sizeof(L.twx_result)), dtype=np.uint8) if fetch else None
        self._check(self._lib.twx_multi_process_recording_dev(self._h, C.cast(ptrs, C.c_void_p), n_windows_total, n_channels, channel, self._band(band), dptr,
                                                              out.ctypes.data_as(C.c_void_p) if fetch else None))
        return out

    def exchange_only(self, records_per_context: int) -> float:
        """The record exchange alone on what the last device-resident call left in the send buffers; returns its wall time in ms."""
        self._check(self._lib.twx_multi_exchange_only(self._h, int(records_per_context)))
        return float(self.info.gather_ms)

    def fetch_gathered(self, i: int, n_records: int) -> np.ndarray:
        """Context ``i``'s copy of the gathered records of the last :meth:`process_dev`."""
        out = np.empty((n_records, C.sizeof(L.twx_result)), dtype=np.uint8)
        self._check(self._lib.twx_multi_fetch_gathered(self._h, int(i), out.ctypes.data_as(C.c_void_p), n_records))
        return out
