"""Acquisition stage of experiments/231001_DLL_PLL/rxcomplex.cpp on the HIP library, with the
program's own arithmetic: x2 FFT-domain input interpolation (``short2double`` :914-963), replica = zero-padded sampled
code (``PRN_sampling`` :965-978, ``memcpy_acq`` :980-987, FFT :434-437), per trial carrier ``downconv_acq`` (:1039-1049)
-> FFT(nfft = 2^k) -> ``cross_spectrum`` with its pass-band mask and 1/n^2 (:1001-1018) -> IFFT -> ``cblas_izamax``
(:553), coarse sweep plus step halving (:534-567), SNR gate (:570-573).

Where the work happens: both interpolation and the sweep are the library's fused FFT chain fed with a replica SPECTRUM
(``twx_set_code_spectrum``) — the interpolation is a 2-phase "correlation" of the raw samples with a weight vector
(zero-padding a spectrum = polyphase inverse transforms), each trial carrier one ``processing(d,df)`` on
the interpolated samples which never leave the device (``twx_xcorr_map_dev`` -> ``twx_caf_freqs_cdev``).  The one-off
replica set-up uses the context's own forward transform (``twx_fft_forward``).  All sample arithmetic is in
libtwstft_hip.so; numpy only builds the small tables.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .correlator import Correlator

NINTERP = 2           # rxcomplex.cpp:29


def _band_mask(n: int, df: float, fmax: float, fmin: float) -> np.ndarray:
    """``idx*df < fmax && idx*df > fmin && idx != 0`` on signed bin indices (rxcomplex.cpp:1007-1008, :1025-1026)."""
    i = np.arange(n)
    idx = np.where(i >= n // 2, i - n, i)
    f = idx.astype(np.float64) * df
    return (f < fmax) & (f > fmin) & (idx != 0)


def prn_sampling(nobs: int, code_pm1, rc: float, fs: float, clen: int, delay_ns: float = 0.0) -> np.ndarray:
    """``PRN_sampling`` rxcomplex.cpp:965-978 (index arithmetic only; no samples involved)."""
    i = np.arange(nobs, dtype=np.float64)
    idx = np.floor(np.fmod((i / fs - delay_ns * 1.0e-9) * float(rc), float(clen))).astype(np.int64)
    idx = np.where(idx < 0, idx + clen, np.where(idx >= clen, idx - clen, idx))
    return np.asarray(code_pm1, dtype=np.float64)[idx]


class Interpolator:
    """``short2double``: one channel of an int16 capture resident on the device -> complex64 at NINTERP x the rate,
    on the device.  n_in = samples per call (5 000 000 for the 1-s buffer of rxcomplex.cpp:469)."""

    def __init__(self, n_in: int, device: int = -1):
        self.n_in, self.nobs = int(n_in), int(n_in) * NINTERP
        self.cor = Correlator(lfsr=(20, 9, self.n_in), fs=1.0, sps=1, Nint=0, device=device, nphase=NINTERP)
        half = self.n_in
        w = np.ones(half, dtype=np.complex128)
        w[half // 2:] = 1.0 / float(half)                        # the upper half alone is divided by nobs/Ninterp (:934-936)
        # prnmap = ifft(zero-padded FFT(x).*w) carries 1/(2*half) = the final /nobs of :957-959; input scale 1/32768 (:922-928)
        w *= 1.0 / 32768.0
        self.cor.set_code_spectrum(w)
        self.cor.set_remove_mean(False)

    def __call__(self, iq_dev: int, out_dev: int, n_channels: int = 2, channel: int = 0):
        L.check(self.cor._lib.twx_xcorr_map_dev(self.cor._h, iq_dev, n_channels, channel, 0.0, out_dev), self.cor._h)

    def close(self):
        self.cor.close()


class Acquisition:
    """One receiver channel of rxcomplex.cpp (``channel_info``): replica set-up at construction (:414-437), then
    :meth:`acquire` per buffer.  ``max_batch``: trial carriers per launch (64: 5.05 ms for the 537-carrier sweep at sdr.param
    sizes against 5.45 ms at 16, ``tools/aux_rates.py acq_batch``)."""

    def __init__(self, code_pm1, rc: float, fs: float, nobs: int, fltmax: float | None = None, dec_a: int = 1, device: int = -1,
                 max_batch: int = 64):
        self.code = np.asarray(code_pm1, dtype=np.float64)
        self.clen = self.code.size
        self.rc, self.fs, self.nobs, self.dec_a = float(rc), float(fs), int(nobs), int(dec_a)
        self.fltmax = float(rc) if fltmax is None else float(fltmax)       # ci[i].fltmax = rc :369
        self.fltmin = -self.fltmax
        nfft = 1
        while True:                                                        # :371-376
            nfft *= 2
            if nfft > self.nobs * 2 // self.dec_a:
                break
        self.nfft = nfft
        self.cor = Correlator(lfsr=(20, 9, nfft), fs=self.fs / self.dec_a, sps=1, Nint=0, device=device, max_batch=max_batch)
        wav_t = prn_sampling(self.nobs, self.code, self.rc, self.fs, self.clen)
        wav_acq = np.zeros(nfft, dtype=np.complex128)
        m = self.nobs // self.dec_a
        wav_acq[:m] = wav_t[:m * self.dec_a:self.dec_a]                     # memcpy_acq :980-987 (before the filter)
        with Correlator(lfsr=(20, 9, nfft), fs=self.fs / self.dec_a, sps=1, Nint=0, device=device, precision="f64") as c64:
            self.wav_acq_f = c64.fft(wav_acq)                               # :434-437, on the device, in fp64 (set-up, once)
        # psbb: power of the low-pass filtered waveform (:422-432); nobs is not a power of two -> its own context
        with Correlator(lfsr=(20, 9, self.nobs), fs=self.fs, sps=1, Nint=0, device=device, nphase=1, precision="f64") as cw:
            mask = _band_mask(self.nobs, self.fs / self.nobs, self.fltmax, self.fltmin)
            filt_spec = np.where(mask, cw.fft(wav_t.astype(np.complex128)) / float(self.nobs), 0.0)
            # Parseval: sum |ifft_unnormalised(S)|^2 = nobs * sum |S|^2
            self.psbb = float(self.nobs * np.sum(np.abs(filt_spec) ** 2)) / float(self.nobs)
        band = _band_mask(nfft, (self.fs / self.dec_a) / nfft, self.fltmax, self.fltmin)
        # cross_spectrum: obs*conj(prn)/n^2 in band (:1001-1018); the sqrt(2) of downconv_acq (:1046-1047) is linear and
        # folded in; one 1/n is the ifft normalisation the library applies to prnmap
        spec = np.where(band, 1.4142135624 * np.conj(self.wav_acq_f) / float(nfft), 0.0)
        self.cor.set_code_spectrum(spec)
        self.cor.set_remove_mean(False)

    def _flags(self) -> int:
        return L.TWX_ACQ_IZAMAX | (L.TWX_ACQ_DEC(self.dec_a) if self.dec_a > 1 else 0)

    def bins(self, smp_dev: int, idx: int, freqs):
        """(pk, pk_idx) of the loop body :543-556 for every trial carrier in ``freqs``; ``smp_dev`` = device pointer to
        the interpolated complex64 stream, ``idx`` = start sample.  ``dec_a`` > 1 (the B210 build, :228-230): every
        ``dec_a``-th sample of the stream is correlated (``smp[i*dec]``, :1046-1047)."""
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        out = (L.twx_result * max(f.size, 1))()
        L.check(self.cor._lib.twx_caf_freqs_cdev(self.cor._h, smp_dev + 8 * int(idx), f.ctypes.data_as(C.c_void_p), f.size,
                                                 self._flags(), C.cast(out, C.c_void_p)), self.cor._h)
        pk = np.array([np.hypot(out[i].xval[0], out[i].xval[1]) for i in range(f.size)])
        return pk, np.array([out[i].indice0 for i in range(f.size)], dtype=np.int64)

    def acquire(self, smp_dev: int, idx: int, fc_init: float, frange: float, fstep: float):
        """The sweep :534-567 in one library call (``twx_acquire_cdev``: coarse sweep, strict ``pk >`` rule, step halving
        until < 1 Hz, bookkeeping between rounds on the device): returns (fc, pk, pt)."""
        r = L.twx_acq_result()
        L.check(self.cor._lib.twx_acquire_cdev(self.cor._h, smp_dev + 8 * int(idx), float(fc_init), float(frange), float(fstep),
                                               self.nobs // self.dec_a, self._flags(), C.byref(r)), self.cor._h)
        self.n_trials = int(r.n_trials)
        return float(r.fc), float(r.pk), int(r.pt)

    def acquire_host_loop(self, smp_dev: int, idx: int, fc_init: float, frange: float, fstep: float):
        """The same sweep with the rounds driven from the host through :meth:`bins` (one synchronisation per round):
        the cross-check of :meth:`acquire`."""
        pk_best, fc, pt = 0.0, float(fc_init), 0
        while True:
            flow, fhigh = fc - frange, fc + frange
            trial = []
            fcc = flow
            while fcc <= fhigh:
                trial.append(fcc)
                fcc += fstep
            pk, pki = self.bins(smp_dev, idx, trial)
            for f, p, i in zip(trial, pk, pki):
                if p > pk_best:
                    fc, pk_best, pt = f, float(p), int(i) % (self.nobs // self.dec_a)
            fstep = fstep / 2.0
            frange = fstep
            if fstep < 1.0:
                break
        return fc, pk_best, pt

    def gate(self, pk: float, px: float, snr_min: float):
        """:570-573."""
        p = 8.0 * pk * pk / self.psbb
        return p, (1.0 + snr_min) * p > snr_min * px

    def close(self):
        self.cor.close()
