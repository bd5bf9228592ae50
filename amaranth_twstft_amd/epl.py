"""Early / prompt / late tracking loop of the Octave DLL/PLL experiment (host twin over the C ABI).

``experiments/230503_100kchips_withcode/gotracking_inv2.m:149-235``: per code period the block is mixed by the NCO
``exp(1j*2*pi*(-freq0+doppler_freq(l))*time)`` (:157-159), cross-correlated with the late / prompt / early replicas
``zl=xcorr(al,xx,MAXLAG)`` ... (:161-163, ``MAXLAG = points_per_code``, :95: every lag of the LINEAR cross-correlation), the three
peaks feed the early-minus-late DLL discriminator (:187) and the arctangent PLL discriminator (:201-203), and the 2nd-order loop
filter updates the Doppler estimate (:209-235).

The three correlations run on the GPU: a correlator context of 2N samples per replica (``twx_set_code_spectrum`` with the spectrum
of the zero-padded replica, claudio convention = ``fft(a).*conj(fft(xx))``, no interpolation) evaluates the linear cross-correlation
of two N-sample sequences as a circular one of length 2N and returns the arg-max and the complex peak — all the loop reads
(``abs(zl(bbl))``, ``zp(bbp)``, ``abs(ze(bbe))``).  Lag k of ``xcorr`` sits at index k + N + 1 of Octave's vector (1-based) and at
circular index k mod 2N of the map.  The loop arithmetic is host Python, as in the script.  UNPINNED (Octave only; oracle:
``oracle.epl_step``).
"""
from __future__ import annotations

import math

import numpy as np

from .correlator import Correlator


def replicas(chips, sps: int = 2):
    """``ap`` = +-1 code held ``sps`` samples, ``al`` = one sample late, ``ae`` = one sample early (:36-41)."""
    ap = np.repeat(2.0 * np.asarray(chips, dtype=np.float64) - 1.0, sps)
    al = np.concatenate([ap[-1:], ap[:-1]])
    ae = np.concatenate([ap[1:], ap[:1]])
    return al, ap, ae


class EplTracker:
    """State of the loop between code periods: ``l`` (1-based block number), ``doppler_freq`` (list, Hz), ``time_end`` (s: the last
    sample time of the previous block, :156), ``code_phase`` / ``carrier_phase`` (the filtered values, :227-228)."""

    def __init__(self, chips, fs: float = 5e6, sps: int = 2, freq0: float = 0.0, T_blk: float = 80e-3, delay_spacing: float = 0.5,
                 B_DLL: float = 2.0, B_PLL: float = 20.0, time_end: float | None = None, precision: str = "f64", device: int = -1):
        self.fs, self.freq0, self.T_blk, self.delay_spacing, self.B_DLL, self.B_PLL = fs, freq0, T_blk, delay_spacing, B_DLL, B_PLL
        self.al, self.ap, self.ae = replicas(chips, sps)
        self.n = self.ap.size                                   # points_per_code
        self.zeta = 1 / math.sqrt(2)                             # :146-147
        self.omega_n = B_PLL / .53
        self.l = 1
        self.doppler_freq = [0.0]                                # doppler_freq=0 (:145)
        self.code_phase = 0.0
        self.carrier_phase = 0.0
        self.time_end = (fs / 10 - 1) / fs if time_end is None else time_end      # time of the 0.1-s acquisition block (:22,53)
        self.history: list[dict] = []
        m = 2 * self.n
        self._cor = []
        for a in (self.al, self.ap, self.ae):
            c = Correlator(lfsr=(20, 9, m), fs=fs, sps=1, Nint=0, convention="claudio", precision=precision, device=device, var_ddof=0)
            pad = np.zeros(m)
            pad[: self.n] = a
            c.set_code_spectrum(np.conj(np.fft.fft(pad)))        # the context multiplies fft(y) by this; claudio: conj of the product
            self._cor.append(c)

    def close(self):
        for c in self._cor:
            c.close()
        self._cor = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _peak(self, cor, xx_pad):
        r = cor.processing_complex(xx_pad, df=0.0)[0]
        i = r.indice                                             # 0-based index of the circular map of 2N lags
        lag = i if i <= self.n else i - 2 * self.n
        return lag + self.n + 1, r.xval                          # Octave's 1-based index into the 2N+1 vector; ifft(fft.*conj(fft)) IS the sum

    def step(self, x) -> dict:
        """One code period: ``x`` = the N complex samples of the tracked channel (:151-155).  Returns the quantities of this block."""
        x = np.asarray(x, dtype=np.complex128).reshape(-1)
        if x.size != self.n:
            raise ValueError("a block is one code period")
        l = self.l
        fd = self.doppler_freq[l - 1]
        time = self.time_end + np.arange(1, x.size + 1) / self.fs                              # :156
        xx = x * np.exp(1j * (2 * np.pi * (-self.freq0 + fd) * time))                         # :157-159
        pad = np.zeros(2 * self.n, dtype=np.complex128)
        pad[: self.n] = xx
        (bbl, zl), (bbp, zp), (bbe, ze) = (self._peak(c, pad) for c in self._cor)              # :161-177
        code_phase_error = self.delay_spacing * (abs(ze) - abs(zl)) / (abs(ze) + abs(zl) + 2 * abs(zp))      # :187
        filtered_code_phase_error = self.T_blk * self.B_DLL / .25 * code_phase_error          # :193
        measured_code_phase = self.code_phase + code_phase_error
        filtered_code_phase = self.code_phase + filtered_code_phase_error
        # Octave: atan(y/0) = +-pi/2 (and NaN for 0/0), where Python's division raises
        ratio = zp.imag / zp.real if zp.real != 0 else (math.copysign(math.inf, zp.imag) if zp.imag != 0 else math.nan)
        delta_theta = math.atan(ratio) / (2 * math.pi)                                          # :201
        sortie = math.atan2(zp.imag, zp.real) / (2 * math.pi)                                   # :202
        doppler_freq_error = self.T_blk / 2 * delta_theta                                       # :204
        filtered_carrier_phase_error = (2 * self.zeta * self.omega_n * self.T_blk - 3 / 2 * self.omega_n ** 2 * self.T_blk ** 2) * delta_theta   # :210
        filtered_doppler_freq_error = self.omega_n ** 2 * self.T_blk * delta_theta              # :211
        measured_carrier_phase = self.carrier_phase + delta_theta
        filtered_carrier_phase = self.carrier_phase + filtered_carrier_phase_error
        measured_doppler_freq = fd + doppler_freq_error
        filtered_doppler_freq = fd + filtered_doppler_freq_error
        self.code_phase, self.carrier_phase = filtered_code_phase, filtered_carrier_phase       # :227-228
        self.time_end = float(time[-1])
        self.l = l + 1
        self.doppler_freq.append(filtered_doppler_freq)                                         # :249
        out = dict(l=l, bbl=bbl, bbp=bbp, bbe=bbe, zl=zl, zp=zp, ze=ze, code_phase_error=code_phase_error, delta_theta=delta_theta, sortie=sortie,
                   measured_code_phase=measured_code_phase, filtered_code_phase=filtered_code_phase, measured_carrier_phase=measured_carrier_phase,
                   filtered_carrier_phase=filtered_carrier_phase, measured_doppler_freq=measured_doppler_freq, doppler_freq=filtered_doppler_freq)
        self.history.append(out)
        return out


class CodeStepTracker:
    """The SECOND DLL experiment, ``experiments/230503_100kchips_withcode/gotracking_test.m:121-187``: late / prompt / early correlations over
    ``MAXLAG`` lags, the coherent early-minus-late discriminator ``d=(|ze|^2-|zl|^2)/(|ze|^2+|zl|^2)`` (:156) that STEPS the code by one
    sample when it leaves (-0.5, 0.5) (:159-168: the prompt replica is rotated, late and early rebuilt from it), and the first-order loop
    of the script's TEST block on the arctangent of the prompt peak (:66-72,:171-180).

    After the first block the script correlates over +-20 lags only (``MAXLAG=20``, :158): that is the DIRECT sliding dot product of the
    tracking stage (``twx_sliding_dot``: the NCO mix of :126-127 inside the kernel, ``ff = (freq0-freq)/fs`` cycles per sample, the phase of
    the block's first sample as ``phi``) — a linear ``xcorr`` of two N-sample sequences is the circular sum over N + 20
    zero-padded samples.  ``z[k] = xcorr(a,xx)[k] = conj(sum_i xx[i] a[i+k])``, i.e. the conjugate of the kernel's lag ``-k``.  The first
    block (every lag) runs on the 2N-sample correlator contexts of :class:`EplTracker`.  Host arithmetic as in the script.  UNPINNED
    (Octave only; oracle: ``oracle.codestep_step``)."""

    NLAG = 20

    def __init__(self, chips, fs: float = 5e6, sps: int = 2, freq0: float = 0.0, coef=(1.0, 0.0, 0.05 / 6, 0.0), time_end: float | None = None,
                 precision: str = "f64", device: int = -1):
        self.fs, self.freq0, self.coef = fs, freq0, tuple(coef)
        self.al, self.ap, self.ae = replicas(chips, sps)
        self.n = self.ap.size
        self._chips, self._sps, self._precision, self._device = chips, sps, precision, device
        self.l, self.maxlag = 1, self.n                           # MAXLAG=points_per_code (:87)
        self.freq = self.freqm1 = self.freqm2 = self.ym1 = 0.0    # :80-84
        self.time_end = (self.n - 1) / fs if time_end is None else time_end      # time=[0:points_per_code-1]'/fs (:44): the last sample of the alignment block
        self.history: list[dict] = []

    def _full_peaks(self, x):
        """First block: every lag, through the 2N-sample contexts (replicas as they stand now)."""
        t = EplTracker(self._chips, fs=self.fs, sps=self._sps, precision=self._precision, device=self._device)
        try:
            time = self.time_end + np.arange(1, x.size + 1) / self.fs
            xx = x * np.exp(1j * (2 * np.pi * (-self.freq0 + self.freq) * time))
            pad = np.zeros(2 * self.n, dtype=np.complex128)
            pad[: self.n] = xx
            return [t._peak(c, pad) for c in t._cor]             # (1-based index into the 2N+1 vector, value)
        finally:
            t.close()

    def _narrow_peaks(self, raw_iq):
        """Later blocks: +-20 lags by the direct sliding dot product, NCO inside the kernel."""
        from . import tracking
        n, nl = self.n, self.NLAG
        m = n + nl
        pad = np.zeros(2 * m, dtype=np.int16)
        pad[: 2 * n] = raw_iq
        ffc = (self.freq0 - self.freq) / self.fs                  # exp(-2 pi j (ff i + phi)) = exp(+2 pi j (-freq0+freq) t_i), t_i = time_end + (i+1)/fs
        phi = ((self.freq0 - self.freq) * (self.time_end + 1.0 / self.fs)) % 1.0
        out = []
        for a in (self.al, self.ap, self.ae):
            rep = np.zeros(m, dtype=np.float32)
            rep[:n] = a
            sd = tracking.sliding_dot(pad, rep, nobs=m, ncodes=1, nlag=nl, ff=ffc, phi=phi, scale=float(m))[0]
            z = np.conj(sd[::-1])                                 # z[k + nl] = xcorr(a, xx, 20)[k]
            i = int(np.abs(z).argmax())
            out.append((i + 1, complex(z[i])))
        return out

    def step(self, raw_iq) -> dict:
        """One code period: ``raw_iq`` = the block's N samples of the tracked channel as int16 ``[I Q I Q ...]`` (the script reads the capture
        raw: no mean is removed, :122-124)."""
        raw_iq = np.ascontiguousarray(raw_iq, dtype=np.int16).reshape(-1)
        if raw_iq.size != 2 * self.n:
            raise ValueError("a block is one code period")
        x = raw_iq[0::2].astype(np.float64) + 1j * raw_iq[1::2].astype(np.float64)
        (bbl, zl), (bbp, zp), (bbe, ze) = self._full_peaks(x) if self.maxlag >= self.n else self._narrow_peaks(raw_iq)
        d = (abs(ze ** 2) - abs(zl) ** 2) / (abs(ze) ** 2 + abs(zl) ** 2)                      # :156
        self.maxlag = self.NLAG                                                                 # :158
        ap = self.ap
        if d < -0.5:                                                                            # :159-163
            ap = np.concatenate([ap[-1:], ap[:-1]])
        if d > 0.5:                                                                             # :164-168
            ap = np.concatenate([ap[1:], ap[:1]])
        if d < -0.5 or d > 0.5:
            self.ap, self.al, self.ae = ap, np.concatenate([ap[-1:], ap[:-1]]), np.concatenate([ap[1:], ap[:1]])
        ratio = zp.imag / zp.real if zp.real != 0 else (math.copysign(math.inf, zp.imag) if zp.imag != 0 else math.nan)
        yp = math.atan(ratio)                                                                   # :172
        yyp = math.atan2(zp.imag, zp.real)                                                      # :173
        freq = self.coef[0] * self.freqm1 + self.coef[1] * self.freqm2 + self.coef[2] * yp + self.coef[3] * self.ym1      # :174
        out = dict(l=self.l, bbl=bbl, bbp=bbp, bbe=bbe, zl=zl, zp=zp, ze=ze, d=float(d), u=abs(zp), yp=yp, yyp=yyp, freq=freq,
                   stepped=int(d > 0.5) - int(d < -0.5))
        self.freqm2, self.freqm1, self.ym1, self.freq = self.freqm1, freq, yp, freq             # :175-178
        self.time_end = self.time_end + self.n / self.fs
        self.l += 1
        self.history.append(out)
        return out
