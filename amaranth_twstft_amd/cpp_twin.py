"""Conventions of the C++ twin ``processing/CPP/main.cpp`` that differ from the Octave/numpy scripts.

* ``file_level_df`` — ``GoRanging::df`` (:363-450): ONE carrier estimate per capture FILE and channel from every
  25th sample of the whole file (mixed by ``foffset``), FFT of the squared series, arg-max (channel 1 inside
  ±2·8 kHz, channel 2 over the whole spectrum), ``freq(pos)/2 + foffset``.  The series length ``file_samples/25`` is
  arbitrary, so the transform is evaluated with Bluestein's identity on the library's own two-pass FFT (three
  device transforms of a 2^a 3^b 5^c length >= 2L-1, fp64 context; ``twx_fft_forward``) — once per file, not a hot path.
  The estimates are what a maintainer of the C++ program passes as ``df`` to ``twx_process_windows`` /
  ``twx_process_file`` (``df_const``).
* the ``<capture>C.mat`` container is ``results_io.save_cpp_mat``.
"""
from __future__ import annotations

import numpy as np

from . import plans
from .correlator import Correlator


MAX_F64_LEN = plans.MAX_N1 * 10000          # no N1 x N2 pair is longer (plans.choose: N1 <= MAX_N1, N2 <= 10000)


def _smooth_candidates(lo: int, hi: int):
    """Even 2^a 3^b 5^c 7^d in [lo, hi], ascending (enumerated, not searched: a handful of thousand numbers at most)."""
    out = []
    p7 = 1
    while p7 <= hi:
        p5 = p7
        while p5 <= hi:
            p3 = p5
            while p3 <= hi:
                v = p3 * 2
                while v <= hi:
                    if v >= lo:
                        out.append(v)
                    v *= 2
                p3 *= 3
            p5 *= 5
        p7 *= 7
    return sorted(out)


def _smooth_len(lo: int) -> int:
    """Smallest even 2^a 3^b 5^c 7^d >= lo for which the plan generator has a COMPLEX-DOUBLE pair (the Bluestein context is
    opened with precision f64).  Fails at once for series the library cannot transform in one piece."""
    if lo > MAX_F64_LEN:
        raise ValueError(f"series too long for the file-level carrier estimate: a transform of >= {lo} points is needed and the "
                         f"longest fp64 plan pair holds {MAX_F64_LEN} (captures over ~{MAX_F64_LEN // 2 * 25 / 5e6:.0f} s at 5 Msps "
                         "and N = 25: estimate the carrier on a part of the file)")
    for m in _smooth_candidates(lo, min(4 * lo + 64, MAX_F64_LEN)):
        if plans.choose(m, f64=True) is not None:
            return m
    raise ValueError(f"no fp64 transform length near {lo} fits the library's plans (series too long)")


class ArbitraryFFT:
    """Length-L DFT (any L) through the library: X[k] = conj(w[k]) * sum_n (x[n] conj(w[n])) w[k-n], w[n] = exp(i pi n^2 / L)
    — a circular convolution of length M >= 2L-1 done with three device FFTs.  Phases from n^2 mod 2L in integers."""

    def __init__(self, L: int, device: int = -1):
        self.L = int(L)
        self.M = _smooth_len(max(2 * self.L - 1, 4000))
        self.cor = Correlator(lfsr=(20, 9, self.M), fs=1.0, sps=1, Nint=0, device=device, precision="f64", max_batch=1)
        n = np.arange(self.L, dtype=np.int64)
        ph = ((n * n) % (2 * self.L)).astype(np.float64) / float(self.L)         # angle / pi, reduced exactly
        self.w = np.cos(np.pi * ph) + 1j * np.sin(np.pi * ph)
        b = np.zeros(self.M, dtype=np.complex128)
        b[:self.L] = self.w
        b[self.M - self.L + 1:] = self.w[1:][::-1]
        self.fb = self.cor.fft(b)

    def __call__(self, x) -> np.ndarray:
        a = np.zeros(self.M, dtype=np.complex128)
        a[:self.L] = np.asarray(x, dtype=np.complex128) * np.conj(self.w)
        prod = self.cor.fft(a) * self.fb
        conv = np.conj(self.cor.fft(np.conj(prod))) / float(self.M)              # inverse transform through the forward one
        return np.conj(self.w) * conv[:self.L]

    def close(self):
        self.cor.close()


def _linspace(start: float, end: float, num: int) -> np.ndarray:
    """``GoRanging::linspace`` main.cpp:734-757: start + delta*i, last element = end."""
    if num <= 1:
        return np.full(num, start, dtype=np.float64)
    f = start + ((end - start) / (num - 1)) * np.arange(num, dtype=np.float64)
    f[-1] = end
    return f


def file_level_df(path: str, fs: float = 5e6, N: int = 25, remote: int = 0, foffset: float = 0.0, device: int = -1):
    """``GoRanging::df`` (processing/CPP/main.cpp:363-450) on a 2-channel int16 capture ``[I1 Q1 I2 Q2]``.
    Returns (foffset1, foffset2 or None)."""
    raw = np.memmap(path, dtype=np.int16, mode="r")
    nrec = raw.size // (4 * N)                                     # file_size :375 (records of N samples)
    # every N-th sample: fread 4 shorts, fseek 4(N-1) :379-382 — a strided view of the mapping: only the rows used are copied
    rec = np.ascontiguousarray(raw[: nrec * N * 4].reshape(nrec * N, 4)[::N])
    t = np.concatenate(([0.0], np.cumsum(np.full(nrec - 1, float(N) / fs))))          # t += N/fs, accumulated :392
    lo = np.exp((-1j * 2.0 * np.float64(np.float32(2.0)) / 2.0 * np.pi * foffset) * t)   # tlo*foffset*t, tlo = -j*2*pi :28,372,386
    out = [None, None]
    eng = ArbitraryFFT(nrec, device)
    try:
        freq = _linspace(-fs / 2 / N, fs / 2 / N, nrec)           # :399
        frange = 8000.0
        kmax = int(np.nonzero(freq < 2 * frange)[0][-1])          # :401-406 (last index satisfying each test)
        kmin = int(np.nonzero(freq <= -2 * frange)[0][-1])
        for ch in ((0,) if remote else (0, 1)):
            dx = rec[:, 2 * ch].astype(np.float64) + 1j * rec[:, 2 * ch + 1].astype(np.float64)
            mean = dx.sum() / float(nrec)                          # mean of the RAW samples :384,416 ...
            x = dx * lo - mean                                     # ... subtracted from the MIXED ones :418
            f = eng(x * x)                                         # :419-421
            sh = np.zeros(nrec, dtype=np.complex128)               # the two memcpy halves :423-424 (an odd length leaves the last element 0)
            h = nrec // 2
            sh[:h] = f[h:2 * h] if nrec % 2 == 0 else f[h:h + h]
            sh[h:2 * h] = f[:h]
            if ch == 0:
                pos = int(np.abs(sh[kmin:kmax]).argmax()) + kmin   # :427-430
            else:
                pos = int(np.abs(sh).argmax())                     # :443 (whole spectrum)
            out[ch] = float(freq[pos] / 2.0 + np.float32(foffset))  # _foffset is a float member :59
    finally:
        eng.close()
    return out[0], out[1]
