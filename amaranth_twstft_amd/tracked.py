"""Tracked multi-code ranging: binding of the library's ``twx_tracked_*`` entry points.

The capture loops of acquisition/claudio_aligned_code_ranging_separate.m:143-205 (``search_df`` :27-47, per-chunk
carrier, the 40-ms code loop with re-alignment, the ``dold`` carry), of its ``_re_`` twin and of
claudio_aligned_code_lo_separate.m:117-164 run in C++ inside libtwstft_hip.so (``csrc/twx_tracked_core.h`` = the
control flow, ``csrc/twx_tracked.hip`` = the device side), so a C, MEX or Octave host reaches them through the same
C ABI (``include/twstft_hip.h``, ``mex/twstft_tracked_mex.cpp``).  This module only marshals arguments and results.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L

MODES = {"ranging": L.TWX_TRK_RANGING, "re": L.TWX_TRK_RE, "lo": L.TWX_TRK_LO}


class _DevBuf:
    """A device allocation with host copies in and out (tests and tools)."""

    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, nbytes
        self.ptr = lib.twx_dev_alloc(nbytes)
        if not self.ptr:
            raise MemoryError("twx_dev_alloc failed")

    def upload(self, offset, arr):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        L.check(self.lib.twx_memcpy_h2d(self.ptr + offset, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, offset, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        L.check(self.lib.twx_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr + offset, nbytes))
        return out

    def close(self):
        if self.ptr:
            self.lib.twx_dev_free(self.ptr)
            self.ptr = None


class TrackedRanging:
    """One code + one GPU + one of the three script flavours (``mode`` = ``ranging`` | ``re`` | ``lo``).

    ``ls_samples`` = ``fs*ls`` complex samples per chunk (:16,148; default 2 s); ``band`` = (lo, hi) Hz overrides the
    mode's search band (``band_hz`` = ±that, kept for the ranging default ±8 kHz :135); ``OP`` = the station flag that
    selects the sign of the remote band (:137-141)."""

    def __init__(self, chips, fs=5e6, sps=2, Nint=1, ls_samples=None, band_hz=None, df_threshold=20.0, device=-1,
                 precision="f32", max_batch=0, mode="ranging", OP=0, band=None):
        self._lib = lib = L.load()
        cfg = L.twx_tracked_config()
        L.check(lib.twx_tracked_defaults(MODES[mode], int(OP), float(fs), C.byref(cfg)))
        self._chips = np.ascontiguousarray(chips, dtype=np.uint8)
        cfg.chips = self._chips.ctypes.data_as(C.POINTER(C.c_uint8))
        cfg.n_chips = self._chips.size
        cfg.sps, cfg.nint = sps, Nint
        if ls_samples is not None:
            cfg.chunk_samples = int(ls_samples)
        if band is not None:
            cfg.band_lo_hz, cfg.band_hi_hz = float(band[0]), float(band[1])
        elif band_hz is not None:
            cfg.band_lo_hz, cfg.band_hi_hz = -float(band_hz), float(band_hz)
        cfg.df_threshold = df_threshold
        cfg.precision = {"f32": L.TWX_F32, "f64": L.TWX_F64}[precision]
        cfg.device, cfg.max_batch = device, max_batch
        h = C.c_void_p()
        rc = lib.twx_tracked_create(C.byref(cfg), C.byref(h))
        if rc == L.TWX_E_SIZE:          # a code length without a built plan pair: build the plug-ins (hipcc), retry
            from . import plans
            plans.ensure(int(cfg.n_chips) * int(sps), int(cfg.precision), lib)
            rc = lib.twx_tracked_create(C.byref(cfg), C.byref(h))
        if rc:
            raise L.TwxError(rc, (lib.twx_tracked_last_error(None) or b"?").decode())
        self._h = h
        self.mode, self.fs, self.Nint = mode, fs, Nint
        self.n = self._chips.size * sps
        self.L = int(cfg.chunk_samples)
        self.default_skip_samples = int(cfg.skip_samples)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.twx_tracked_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise L.TwxError(rc, (self._lib.twx_tracked_last_error(self._h) or b"?").decode())

    # -- search_df (:27-47) ----------------------------------------------------------------
    def search_df(self, chunk_i16) -> int:
        """0-based index into the shifted axis of the accepted carrier bin, -1 if none."""
        chunk = np.ascontiguousarray(chunk_i16, dtype=np.int16).reshape(-1)
        kb = C.c_int64()
        self._check(self._lib.twx_tracked_search_df(self._h, chunk.ctypes.data_as(C.c_void_p), chunk.size // 2, C.byref(kb)))
        return int(kb.value)

    # -- the loop (:143-205) ---------------------------------------------------------------
    def _collect(self, s) -> dict:
        codes = (L.twx_tracked_code * max(s.n_codes, 1))()
        df = np.zeros(max(s.n_chunks, 1))
        moved = np.zeros(max(s.n_moved, 1), dtype=np.int64)
        mv = np.zeros(max(s.n_moved, 1))
        self._check(self._lib.twx_tracked_fetch(self._h, C.cast(codes, C.c_void_p), df.ctypes.data_as(C.c_void_p),
                                                moved.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p)))
        rec = np.frombuffer(bytes(codes), dtype=np.float64).reshape(-1, 7)[:s.n_codes]
        return dict(xval=list(rec[:, 0] + 1j * rec[:, 1]), indice1=list(rec[:, 2]), correction1=list(rec[:, 3]),
                    SNR1r=list(rec[:, 4]), SNR1i=list(rec[:, 5]), puissance1=list(rec[:, 6]), df=list(df[:s.n_chunks]),
                    moved=[int(v) for v in moved[:s.n_moved]], movedval=list(mv[:s.n_moved]), kbon=int(s.kbon),
                    batches=int(s.batches), puissancecode=s.puissancecode, puissancenoise=s.puissancenoise)

    def run(self, raw, skip_samples: int = 0, kbon: int | None = None) -> dict:
        """Capture held in host memory (int16 ``[I Q]…``).  Returns the script's per-code vectors as lists."""
        raw = np.ascontiguousarray(np.asarray(raw).reshape(-1), dtype=np.int16)
        s = L.twx_tracked_summary()
        self._check(self._lib.twx_tracked_host(self._h, raw.ctypes.data_as(C.c_void_p), raw.size // 2, int(skip_samples),
                                               -1 if kbon is None else int(kbon), C.byref(s)))
        return self._collect(s)

    def run_file(self, path: str, skip_seconds: float | None = None, kbon: int | None = None) -> dict:
        """Whole single-channel sc16 capture file; ``skip_seconds`` None = the mode's own skip (30 s, ``lo``: none, :128)."""
        s = L.twx_tracked_summary()
        skip = -1 if skip_seconds is None else int(skip_seconds * self.fs)
        self._check(self._lib.twx_tracked_file(self._h, os.fsencode(path), skip, -1 if kbon is None else int(kbon), C.byref(s)))
        return self._collect(s)

    STAGES = ("chunk wait", "carrier bins", "band spectrum", "code measurements", "search_df candidates", "tail carry", "whole run")

    def timing(self) -> dict:
        """Wall time of the last run inside each device operation of the control flow (``twx_tracked_timing``): name -> (s, calls)."""
        sec = (C.c_double * len(self.STAGES))()
        calls = (C.c_int64 * len(self.STAGES))()
        self._check(self._lib.twx_tracked_timing(self._h, sec, calls))
        return {n: (float(sec[i]), int(calls[i])) for i, n in enumerate(self.STAGES)}
