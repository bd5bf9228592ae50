"""Host-side mirror of the reference's correlation interface, running on the HIP library.

Names and argument meaning follow the reference scripts so the parity tests read like the
reference's own usage:

* ``Correlator.processing(raw, k)``  ↔  ``processing(d,k)``  processing/Octave/godual_ranging.m:12
  (the window is handed over as raw int16 IQ; mean removal of :80 happens on the device)
* ``Correlator.processing_df(raw, df)``  ↔  ``processing(d,df)``
  acquisition/claudio_aligned_code_ranging_separate.m:49
* ``Correlator.ranging(raw, ...)``  ↔  the window loop ``ranging(filename, prn_code, …)`` of
  experiments/221219_twoway/processing/godual_ranging.py:67-139 / godual_ranging.m:59-102
* ``freq_axis`` / ``band_godual`` / ``band_numpy``  ↔  godual_ranging.m:73,83-89, godual_ranging.py:79-81

All arithmetic on samples happens in libtwstft_hip.so (no numpy/torch fallback).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


def freq_axis(fs: float, n: int) -> np.ndarray:
    """``freq=linspace(-fs/2,fs/2,length(code))`` (godual_ranging.m:73)."""
    return np.linspace(-fs / 2, fs / 2, num=n, dtype=float)


def _band_range(mask_idx: np.ndarray) -> tuple[int, int]:
    if mask_idx.size == 0:
        raise ValueError("empty search band")
    lo, hi = int(mask_idx[0]), int(mask_idx[-1])
    if hi - lo + 1 != mask_idx.size:
        raise ValueError("search band must be contiguous")
    return lo, hi


def band_godual(fs: float, n: int, remote: int = 0, OP: int = 0) -> tuple[int, int]:
    """Search band of godual_ranging.m:83-89 as an inclusive 0-based index range."""
    f = freq_axis(fs, n)
    if remote != 1:
        k = np.nonzero((f < 20000) & (f > -20000))[0]
    elif OP == 1:
        k = np.nonzero((f > -120000) & (f < -80000))[0]
    else:
        k = np.nonzero((f < 120000) & (f > 80000))[0]
    return _band_range(k)


def band_numpy(fs: float, n: int, foffset: float = 0.0, frange: float = 8000.0) -> tuple[int, int]:
    """Search band of experiments/221219_twoway/processing/godual_ranging.py:80-81."""
    f = freq_axis(fs, n)
    return _band_range(np.nonzero((f < 2 * (foffset + frange)) & (f > 2 * (foffset - frange)))[0])


@dataclass
class WindowResult:
    """Outputs of ``processing`` for one channel-window (0-based ``indice``; Octave's is +1)."""
    indice: int
    correction: float
    xval: complex
    xvalm1: complex
    xvalp1: complex
    zwin: np.ndarray
    df: float
    df_index: int
    SNRr: float
    SNRi: float
    puissance: float
    puissancecode: float
    puissancenoise: float
    status: int = 0          # TWX_STATUS_* bits (self-check failed, resampled window all NaN)
    dt: int = 0              # velocity-compensated window: the carried whole-sample offset the script adds to indice (godual_ranging_OP_vitesse.m:68)

    def correction_polyfit(self, half_width: int) -> float:
        """Parabola through 2·half_width+1 magnitude samples around the peak, ``[u,v]=polyfit([-h:+h]',abs(prnmap(indice-h:
        indice+h)),2); -u(2)/2/u(1)`` — ``correction1_1/_2/_3`` of experiments/221207_twoway_codes/processing/
        godual_ranging.m:73-78 (half_width 1, 2, 3), from the seven peak samples the device returns in ``zwin``."""
        if not 1 <= half_width <= 3:
            raise ValueError("half_width must be 1, 2 or 3")
        x = np.arange(-half_width, half_width + 1, dtype=float)
        y = np.abs(self.zwin[3 - half_width:3 + half_width + 1])
        u = np.polyfit(x, y, 2)
        return float(-u[1] / 2 / u[0])

    def delay(self, fs: float, nint: int, sign: int = +1) -> float:
        """``(indice-1±correction)/fs/(2*Nint+1)`` as printed by godual_ranging.m:96."""
        return (self.indice + sign * self.correction) / fs / (2 * nint + 1)


def _to_result(r: L.twx_result) -> WindowResult:
    z = np.array([[p[0], p[1]] for p in r.zwin])
    return WindowResult(int(r.indice0), r.correction, complex(*r.xval), complex(*r.xvalm1), complex(*r.xvalp1),
                        z[:, 0] + 1j * z[:, 1], r.df, int(r.df_index), r.SNRr, r.SNRi, r.puissance,
                        r.puissancecode, r.puissancenoise, int(r.status), int(r.dt))


ALL_CHANNELS = -1


class Correlator:
    """One code + one GPU. Mirrors the globals ``fs Nint code fcode`` of godual_ranging.m:3-5,62-66."""

    def __init__(self, chips=None, fs: float = 5e6, sps: int = 2, Nint: int = 1, *, lfsr: tuple[int, int, int] | None = None,
                 precision: str = "f32", var_ddof: int = 0, snr_rot: int = -1, window: str = "none", convention: str = "godual",
                 device: int = -1, max_batch: int = 0, profile: bool = False, fine_freq: bool = False,
                 chips_q=None, code_levels: str = "bipolar", code_zero_mean: bool = False, nphase: int = 0):
        self._lib = L.load()
        cfg = L.twx_config()
        cfg.fs, cfg.sps, cfg.nint = fs, sps, Nint
        if chips is not None:
            self._chips = np.ascontiguousarray(chips, dtype=np.uint8)
            cfg.chips = self._chips.ctypes.data_as(C.POINTER(C.c_uint8))
            cfg.n_chips = self._chips.size
        elif lfsr is not None:
            cfg.chips = None
            cfg.lfsr_bitlen, cfg.lfsr_taps, cfg.n_chips = lfsr
        else:
            raise ValueError("give chips or lfsr=(bitlen, taps, noiselen)")
        cfg.convention = {"godual": L.TWX_CONV_GODUAL, "claudio": L.TWX_CONV_CLAUDIO}[convention]
        cfg.window = {"none": L.TWX_WIN_NONE, "hamming": L.TWX_WIN_HAMMING}[window]
        cfg.precision = {"f32": L.TWX_F32, "f64": L.TWX_F64}[precision]
        cfg.var_ddof, cfg.snr_rot, cfg.device, cfg.max_batch = var_ddof, snr_rot, device, max_batch
        cfg.flags = ((L.TWX_FLAG_PROFILE if profile else 0) | (L.TWX_FLAG_FINE_FREQ if fine_freq else 0)
                     | (L.TWX_FLAG_CODE_ZERO_MEAN if code_zero_mean else 0))
        # replica variants of the experiment scripts (220616_Besancon/godual.m:5-7, 220822_qpsk_vs_bpsk/goqpsk.m:10-14)
        cfg.code_levels = {"bipolar": L.TWX_CODE_BIPOLAR, "unipolar": L.TWX_CODE_UNIPOLAR}[code_levels]
        cfg.nphase = int(nphase)          # 0 = 2*Nint+1 output phases
        if chips_q is not None:
            self._chips_q = np.ascontiguousarray(chips_q, dtype=np.uint8)
            if chips is None or self._chips_q.size != self._chips.size:
                raise ValueError("chips_q needs chips of the same length")
            cfg.chips_q = self._chips_q.ctypes.data_as(C.POINTER(C.c_uint8))
        h = C.c_void_p()
        rc = self._lib.twx_create(C.byref(cfg), C.byref(h))
        if rc == L.TWX_E_SIZE:
            # a window length the library has no plan pair for: build and load plan plug-ins (hipcc), then retry
            from . import plans
            plans.ensure(int(cfg.n_chips) * int(sps), int(cfg.precision), self._lib)
            rc = self._lib.twx_create(C.byref(cfg), C.byref(h))
        L.check(rc)
        self._h = h
        info = L.twx_info()
        L.check(self._lib.twx_get_info(self._h, C.byref(info)), self._h)
        self.info = info
        self.fs, self.sps, self.Nint = fs, sps, Nint
        self.n = int(info.n)

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.twx_destroy(self._h)
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

    # -- the kernel ----------------------------------------------------------------------
    def _windows(self, raw, n_channels):
        raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1)
        per = self.n * 2 * n_channels
        nwin = raw.size // per          # short final window is dropped (godual_ranging.m:81,102)
        return raw, nwin

    def process(self, raw, n_channels=1, channel=0, band=None, df=None):
        """Run ``processing`` over every full window of an interleaved int16 capture (host memory).
        ``channel = -1`` (``ALL_CHANNELS``): every channel from one upload of the capture → ``{c: [WindowResult …]}``
        (``df``, if given, broadcastable to [nwin, n_channels])."""
        raw, nwin = self._windows(raw, n_channels)
        allch = channel < 0
        nrec = nwin * (n_channels if allch else 1)
        out = (L.twx_result * max(nrec, 1))()
        bptr = None
        dptr = None
        if band is not None:
            b = L.twx_band(int(band[0]), int(band[1]))
            bptr = C.byref(b)
        else:
            if df is None:
                raise ValueError("give band (estimate df) or df (per window)")
            shape = (nwin, n_channels) if allch else (nwin,)
            dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), shape))
            dptr = dfa.ctypes.data_as(C.c_void_p)
        L.check(self._lib.twx_process_windows(self._h, raw.ctypes.data_as(C.c_void_p), nwin, n_channels, channel,
                                              bptr, dptr, C.cast(out, C.c_void_p)), self._h)
        if allch:
            return {c: [_to_result(out[w * n_channels + c]) for w in range(nwin)] for c in range(n_channels)}
        return [_to_result(out[i]) for i in range(nwin)]

    def process_file(self, path: str, n_channels=1, channel=0, band=None, df=None, skip_samples: int = 0,
                     max_windows: int | None = None, raw_records: bool = False):
        """Window loop over a capture file (godual_ranging.m:70-103), pinned double-buffered ingest.
        ``channel = -1``: every channel from one pass over the file → ``{c: [WindowResult …]}``.
        ``raw_records``: return the ``twx_result`` records as a uint8 array [n_records, sizeof(twx_result)] (record
        w*n_channels + c in the all-channel mode) — what the ranks of a multi-GPU job exchange (dist.gather_results)."""
        import os
        per = self.n * 4 * n_channels
        avail = max(0, (os.path.getsize(path) - skip_samples * 4 * n_channels)) // per
        nmax = avail if max_windows is None else min(avail, max_windows)
        allch = channel < 0
        nch_out = n_channels if allch else 1
        out = (L.twx_result * max(nmax * nch_out, 1))()
        bptr = None
        if band is not None:
            b = L.twx_band(int(band[0]), int(band[1]))
            bptr = C.byref(b)
        elif df is None:
            raise ValueError("give band (estimate df) or df")
        ndone = C.c_int64()
        L.check(self._lib.twx_process_file(self._h, os.fsencode(path), n_channels, channel, skip_samples, bptr,
                                           float(df) if df is not None else 0.0, C.cast(out, C.c_void_p), nmax, C.byref(ndone)), self._h)
        if raw_records:
            nrec = ndone.value * nch_out
            return np.frombuffer(bytes(out), dtype=np.uint8).reshape(-1, C.sizeof(L.twx_result))[:nrec].copy()
        if allch:
            return {c: [_to_result(out[w * n_channels + c]) for w in range(ndone.value)] for c in range(n_channels)}
        return [_to_result(out[i]) for i in range(ndone.value)]

    def processing_complex(self, d, k=None, df=None):
        """The reference's own call: ``processing(d,k)`` (godual_ranging.m:12) or ``processing(d,df)``
        (claudio_aligned_code_ranging_separate.m:49) on the complex column ``d`` the scripts build (mean already removed
        by the caller, :80).  ``d``: complex128, a whole number of windows; ``k``: (k_lo, k_hi) or an index array;
        ``df``: scalar or one value per window.  Returns one WindowResult per window."""
        d = np.ascontiguousarray(d, dtype=np.complex128).reshape(-1)
        nwin = d.size // self.n
        if nwin * self.n != d.size:
            raise ValueError("d must hold a whole number of windows")
        out = (L.twx_result * max(nwin, 1))()
        bptr = dptr = None
        if k is not None:
            if not (isinstance(k, tuple) and len(k) == 2):
                k = _band_range(np.asarray(k))
            b = L.twx_band(int(k[0]), int(k[1]))
            bptr = C.byref(b)
        elif df is not None:
            dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), (nwin,)))
            dptr = dfa.ctypes.data_as(C.c_void_p)
        else:
            raise ValueError("give k (band) or df")
        base = d.ctypes.data
        L.check(self._lib.twx_process_complex(self._h, base, base + 8, 2, nwin, bptr, dptr, C.cast(out, C.c_void_p)), self._h)
        return [_to_result(out[i]) for i in range(nwin)]

    def processing(self, raw_window, k, n_channels=1, channel=0) -> WindowResult:
        """``processing(d,k)`` (godual_ranging.m:12): ``k`` = (k_lo, k_hi) or an index array."""
        if not (isinstance(k, tuple) and len(k) == 2):
            k = _band_range(np.asarray(k))
        return self.process(raw_window, n_channels, channel, band=k)[0]

    def processing_df(self, raw_window, df, n_channels=1, channel=0) -> WindowResult:
        """``processing(d,df)`` (claudio_aligned_code_ranging_separate.m:49), godual peak convention."""
        return self.process(raw_window, n_channels, channel, df=df)[0]

    def ranging(self, raw, n_channels=2, channels=(0, 1), band=None, remote=0, OP=0):
        """Window loop of godual_ranging.m:75-102: {channel: [WindowResult …]}."""
        if band is None:
            band = band_godual(self.fs, self.n, remote, OP)
        if tuple(channels) == tuple(range(n_channels)) and 1 < n_channels <= 4:
            return self.process(raw, n_channels, ALL_CHANNELS, band=band)        # one upload for both channels (:91,95)
        return {c: self.process(raw, n_channels, c, band=band) for c in channels}

    # -- delay x Doppler search (experiments/231001_DLL_PLL/rxcomplex.cpp:521-572) -----------------
    def caf_bins(self, raw_window, k_lo: int, k_hi: int, n_channels=1, channel=0):
        """Per-bin (peak |xcorr|, lag) on the integer-bin Doppler grid f = k*fs/N, k = k_lo..k_hi."""
        raw = np.ascontiguousarray(raw_window, dtype=np.int16).reshape(-1)
        assert raw.size >= self.n * 2 * n_channels
        nb = int(k_hi) - int(k_lo) + 1
        pk = np.empty(nb, dtype=np.float64)
        lag = np.empty(nb, dtype=np.int64)
        L.check(self._lib.twx_caf_bins(self._h, raw.ctypes.data_as(C.c_void_p), n_channels, channel, int(k_lo), int(k_hi),
                                       pk.ctypes.data_as(C.c_void_p), lag.ctypes.data_as(C.c_void_p)), self._h)
        return pk, lag

    def caf_bins_dev(self, iq_dev: int, k_lo: int, k_hi: int, n_channels=1, channel=0):
        """:meth:`caf_bins` on a window that already sits in device memory."""
        nb = int(k_hi) - int(k_lo) + 1
        pk = np.empty(nb, dtype=np.float64)
        lag = np.empty(nb, dtype=np.int64)
        L.check(self._lib.twx_caf_bins_dev(self._h, iq_dev, n_channels, channel, int(k_lo), int(k_hi),
                                           pk.ctypes.data_as(C.c_void_p), lag.ctypes.data_as(C.c_void_p)), self._h)
        return pk, lag

    def caf_freqs(self, raw_window, freqs, n_channels=1, channel=0) -> list[WindowResult]:
        """processing(d,df) of the same window for every trial offset in ``freqs`` (Hz)."""
        raw = np.ascontiguousarray(raw_window, dtype=np.int16).reshape(-1)
        assert raw.size >= self.n * 2 * n_channels
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        out = (L.twx_result * max(f.size, 1))()
        L.check(self._lib.twx_caf_freqs(self._h, raw.ctypes.data_as(C.c_void_p), n_channels, channel,
                                        f.ctypes.data_as(C.c_void_p), f.size, C.cast(out, C.c_void_p)), self._h)
        return [_to_result(out[i]) for i in range(f.size)]

    def acquire(self, raw_window, fc_init: float, frange: float, fstep: float, n_channels=1, channel=0):
        """Coarse-to-fine carrier/code-phase acquisition, the loop of rxcomplex.cpp:534-567: sweep
        fc±frange in fstep, keep the highest peak, then halve the step (range = step) until < 1 Hz.
        Returns (fc, peak magnitude, lag in samples of the interpolated grid)."""
        fc, best_pk, best_lag = float(fc_init), 0.0, 0
        while True:
            flow, fhigh = fc - frange, fc + frange
            trial = np.arange(flow, fhigh + 1e-9 * max(1.0, abs(fhigh)), fstep)
            res = self.caf_freqs(raw_window, trial, n_channels, channel)
            for f, r in zip(trial, res):
                pk = abs(r.xval)
                if pk > best_pk:
                    fc, best_pk, best_lag = float(f), pk, r.indice
            fstep = fstep / 2.0
            frange = fstep
            if fstep < 1.0:
                break
        return fc, best_pk, best_lag

    @staticmethod
    def acquisition_gate(pk: float, px: float, snr_min: float, psbb: float = 1.0):
        """Lock decision after :meth:`acquire`: peak signal power ``8*pk^2/psbb`` against the total received power ``px``,
        ``(1+snr_min)*pk_power > snr_min*px`` (rxcomplex.cpp:570-573).  Returns (locked, pk_power)."""
        pk_power = 8.0 * pk * pk / psbb
        return (1.0 + snr_min) * pk_power > snr_min * px, pk_power

    # -- inspection ----------------------------------------------------------------------
    def fft(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.complex128)
        assert x.size == self.n
        out = np.empty(self.n, dtype=np.complex128)
        L.check(self._lib.twx_fft_forward(self._h, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), self._h)
        return out

    def code_spectrum(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.complex128)
        L.check(self._lib.twx_get_code_spectrum(self._h, out.ctypes.data_as(C.c_void_p)), self._h)
        return out

    def set_code_spectrum(self, spec):
        """Replace the replica spectrum the context multiplies ``fft(y)`` with (``conj(fft(code))`` by default) by
        ``spec`` (N complex values, natural FFT order) — twx_set_code_spectrum."""
        sp = np.ascontiguousarray(spec, dtype=np.complex128).reshape(-1)
        if sp.size != self.n:
            raise ValueError("spectrum length must equal the window length")
        L.check(self._lib.twx_set_code_spectrum(self._h, sp.ctypes.data_as(C.c_void_p)), self._h)

    def set_resample(self, vitesse: float, t0: float = 0.0, dt: int = 0):
        """The velocity-compensated window of experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m (:4 ``vitesse=-3.25e-9``): every
        later window of this context is resampled after the NCO mix, ``yi=interp1([0:N-1],y,[0:N-1]*1/(1-vitesse)+t0)`` (:40), with
        ``t0`` / ``dt`` carried from window to window as the script carries them (:41,:68-71; they start at 0).  ``vitesse = 0``: off.
        Each ``WindowResult`` then holds its window's ``dt``: the script's ``indice1(p)`` is ``indice + 1 + dt``."""
        L.check(self._lib.twx_set_resample(self._h, float(vitesse), float(t0), int(dt)), self._h)

    def get_resample(self) -> tuple[float, float, int]:
        """(vitesse, t0, dt) as they stand before the next window."""
        v, t0, dt = C.c_double(), C.c_double(), C.c_int64()
        L.check(self._lib.twx_get_resample(self._h, C.byref(v), C.byref(t0), C.byref(dt)), self._h)
        return v.value, t0.value, int(dt.value)

    def set_snr_estimators(self, bruit_len: int = 0, noise_square_len: int = 0):
        """The other SNR estimators of experiments/220830_OP/process_OP.m beside the wipe-off SNR: ``bruit`` =
        ``var(prnmap(indice+20:indice+20+bruit_len-1))`` (:119-121: 1001, :138: 10001) and ``valmax_square`` / ``noise_square`` =
        the carrier peak of ``fftshift(abs(fft(d1.^2)))`` and the variance of the ``noise_square_len`` bins from 20 above it (:95-97:
        10001).  0 = off.  :meth:`snr_estimators` returns them for the records of the last call."""
        L.check(self._lib.twx_set_option(self._h, L.TWX_OPT_BRUIT_LEN, int(bruit_len)), self._h)
        L.check(self._lib.twx_set_option(self._h, L.TWX_OPT_NOISE_SQUARE_LEN, int(noise_square_len)), self._h)

    def snr_estimators(self, n_records: int) -> list[dict]:
        """``[{bruit, valmax_square, noise_square}, ...]`` for the first ``n_records`` records of the last call (same order)."""
        out = (L.twx_extra * max(n_records, 1))()
        L.check(self._lib.twx_fetch_extra(self._h, out, n_records), self._h)
        return [dict(bruit=out[i].bruit, valmax_square=out[i].valmax_square, noise_square=out[i].noise_square) for i in range(n_records)]

    def set_selfcheck(self, value: int = 1):
        """``TWX_OPT_SELFCHECK``: Parseval's identity per row of the fused middle pass; a window with a row outside the tolerance comes
        back with ``status & TWX_STATUS_SELFCHECK``.  0 = off, 1 = 1e-5 relative, > 1 = the tolerance in units of 1e-9."""
        L.check(self._lib.twx_set_option(self._h, L.TWX_OPT_SELFCHECK, int(value)), self._h)

    def selfcheck_stats(self, reset: bool = True) -> tuple[float, int]:
        """(largest relative Parseval deviation seen, rows flagged) since the last reset."""
        d, n = C.c_double(), C.c_int64()
        L.check(self._lib.twx_selfcheck_stats(self._h, C.byref(d), C.byref(n), 1 if reset else 0), self._h)
        return d.value, int(n.value)

    def set_remove_mean(self, on: bool):
        """``d=d-mean(d)`` before the NCO (default on; the reference's callers do it, godual_ranging.m:80,94)."""
        L.check(self._lib.twx_set_option(self._h, L.TWX_OPT_REMOVE_MEAN, 1 if on else 0), self._h)

    def xcorr_map(self, raw_window, df, n_channels=1, channel=0, raw_mean: bool = False) -> np.ndarray:
        """The whole interpolated ``prnmap`` of one window (``raw_mean``: skip the mean removal)."""
        raw = np.ascontiguousarray(raw_window, dtype=np.int16).reshape(-1)
        assert raw.size >= self.n * 2 * n_channels
        out = np.empty(self.n * (2 * self.Nint + 1), dtype=np.complex128)
        if raw_mean:
            self.set_remove_mean(False)
        try:
            L.check(self._lib.twx_xcorr_map(self._h, raw.ctypes.data_as(C.c_void_p), n_channels, channel, float(df),
                                            out.ctypes.data_as(C.c_void_p)), self._h)
        finally:
            if raw_mean:
                self.set_remove_mean(True)
        return out

    # -- device-resident input (pointers from twx_dev_alloc or any HIP allocation) -------------
    def process_dev(self, iq_dev: int, nwin: int, n_channels=1, channel=0, band=None, df=None):
        """``process`` on ``nwin`` consecutive windows that already sit in device memory at ``iq_dev``
        (``channel = -1``: all channels → ``{c: [WindowResult …]}``)."""
        allch = channel < 0
        nrec = max(nwin * (n_channels if allch else 1), 1)
        nbytes = C.sizeof(L.twx_result) * nrec
        res = self._lib.twx_dev_alloc(nbytes)
        if not res:
            raise MemoryError("twx_dev_alloc failed")
        try:
            bptr = dptr = None
            if band is not None:
                b = L.twx_band(int(band[0]), int(band[1]))
                bptr = C.byref(b)
            else:
                shape = (nwin, n_channels) if allch else (nwin,)
                dfa = np.ascontiguousarray(np.broadcast_to(np.asarray(df, dtype=np.float64), shape))
                dptr = dfa.ctypes.data_as(C.c_void_p)
            L.check(self._lib.twx_process_windows_dev(self._h, iq_dev, nwin, n_channels, channel, bptr, dptr, res), self._h)
            L.check(self._lib.twx_synchronize(self._h), self._h)
            out = (L.twx_result * nrec)()
            L.check(self._lib.twx_memcpy_d2h(C.cast(out, C.c_void_p), res, nbytes))
        finally:
            self._lib.twx_dev_free(res)
        if allch:
            return {c: [_to_result(out[w * n_channels + c]) for w in range(nwin)] for c in range(n_channels)}
        return [_to_result(out[i]) for i in range(nwin)]

    # -- front end and short-code direct path on device-resident data (context's stream, no host round trip) ----
    def fir_decimate_dev(self, iq_dev: int, n_in: int, taps, dec: int, out_i16_dev: int | None = None,
                         out_f32_dev: int | None = None, n_channels=1, channel=0) -> int:
        """``y[m] = sum_j taps[j]*x[m*dec+j]`` of a device-resident capture into device buffers (int16 ``[I Q]`` and/or
        complex64); returns the number of outputs.  The int16 output feeds :meth:`process_dev` directly
        (BASELINE.json configs[4]: 70 Msps → 5 Msps → correlator).  Asynchronous on the context's stream."""
        t = np.ascontiguousarray(taps, dtype=np.float32)
        nout = C.c_int64()
        L.check(self._lib.twx_fir_decimate_dev(self._h, iq_dev, int(n_in), n_channels, channel, t.ctypes.data_as(C.c_void_p), t.size,
                                               int(dec), out_i16_dev, out_f32_dev, C.byref(nout)), self._h)
        return int(nout.value)

    def sliding_dot_dev(self, iq_dev: int, n_samples: int, replica_dev: int, nobs: int, ncodes: int, nlag: int, out_dev: int,
                        pt: int = 0, ff: float = 0.0, phi: float = 0.0, scale: float = 1.0, n_channels=1, channel=0) -> None:
        """±nlag sliding dot products per code period (tracking.sliding_dot) on device-resident samples and replica;
        ``out_dev``: ncodes*(2*nlag+1) complex128 on the device.  Asynchronous on the context's stream."""
        L.check(self._lib.twx_sliding_dot_dev(self._h, iq_dev, int(n_samples), n_channels, channel, int(pt), int(nobs), int(ncodes),
                                              int(nlag), replica_dev, float(ff), float(phi), float(scale), out_dev), self._h)

    def synchronize(self):
        L.check(self._lib.twx_synchronize(self._h), self._h)

    def sqspec_bins_dev(self, iq_dev: int, n_samples: int, bins, n_channels=1, channel=0) -> np.ndarray:
        """``fft(d.^2)`` of an ``n_samples`` chunk at the given signed DFT bins (complex128)."""
        b = np.ascontiguousarray(bins, dtype=np.int64)
        out = np.empty(2 * b.size, dtype=np.float64)
        L.check(self._lib.twx_sqspec_bins_dev(self._h, iq_dev, int(n_samples), n_channels, channel,
                                              b.ctypes.data_as(C.c_void_p), b.size, out.ctypes.data_as(C.c_void_p)), self._h)
        return out[0::2] + 1j * out[1::2]

    def sqspec_band_dev(self, iq_dev: int, n_samples: int, k_lo: int, n_bins: int, n_channels=1, channel=0) -> np.ndarray:
        """``abs(fft(d.^2))`` over ``n_bins`` consecutive signed bins from ``k_lo``; ``n_samples`` a multiple of N."""
        out = np.empty(int(n_bins), dtype=np.float64)
        L.check(self._lib.twx_sqspec_band_dev(self._h, iq_dev, int(n_samples), n_channels, channel, int(k_lo), int(n_bins),
                                              out.ctypes.data_as(C.c_void_p)), self._h)
        return out

    def profile(self, reset=False) -> dict:
        ents = (L.twx_prof_entry * L.TWX_PROF_MAX)()
        n = C.c_int32()
        L.check(self._lib.twx_profile_get(self._h, ents, L.TWX_PROF_MAX, C.byref(n)), self._h)
        out = {ents[i].name.decode(): dict(ms_total=ents[i].ms_total, launches=ents[i].launches, units=ents[i].units)
               for i in range(n.value)}
        if reset:
            L.check(self._lib.twx_profile_reset(self._h), self._h)
        return out


def lfsr_chips_device(bitlen: int, taps: int, n: int) -> np.ndarray:
    """LFSR chips generated by the device kernel (same bytes as prn.lfsr_chips)."""
    lib = L.load()
    out = np.empty(n, dtype=np.uint8)
    L.check(lib.twx_lfsr_chips(bitlen, taps, n, out.ctypes.data_as(C.c_void_p)))
    return out
