"""Short-code tracking correlator: ±nlag sliding dot products per code period on the GPU, then the
reference's power / phase / high-resolution-correlator arithmetic on the handful of results.

Mirrors the tracking branch of experiments/231001_DLL_PLL/rxcomplex.cpp:593-661:
``downconv_trk`` (:1051) + ``cblas_dgemm`` (:605) → ``twx_sliding_dot``;
``get_cor_and_phi`` (:1063-1072) and the HRC delay (:648-661) are O(nlag) host arithmetic.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def sliding_dot(raw, replica, nobs: int, ncodes: int, nlag: int, pt: int = 0, ff: float = 0.0, phi: float = 0.0,
                scale: float = 1.0, n_channels: int = 1, channel: int = 0) -> np.ndarray:
    """complex128 [ncodes, 2*nlag+1]: (scale/nobs)·Σ_i x[pt+p·nobs+i]·e^{-2πj(ff·(p·nobs+i)+phi)}·replica[(i-lag) mod nobs]."""
    lib = L.load()
    raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1)
    n_samples = raw.size // (2 * n_channels)
    rep = np.ascontiguousarray(replica, dtype=np.float32)
    if rep.size != nobs:
        raise ValueError("replica must have nobs entries")
    out = np.empty((ncodes, 2 * nlag + 1), dtype=np.complex128)
    L.check(lib.twx_sliding_dot(raw.ctypes.data_as(C.c_void_p), n_samples, n_channels, channel, pt, nobs, ncodes, nlag,
                                rep.ctypes.data_as(C.c_void_p), ff, phi, scale, out.ctypes.data_as(C.c_void_p)))
    return out


def get_cor_and_phi(res: np.ndarray):
    """``cor = re²+im²``, ``phi = atan2(im,re)/2π`` (rxcomplex.cpp:1063-1072)."""
    return res.real ** 2 + res.imag ** 2, np.arctan2(res.imag, res.real) / (2 * np.pi)


def hrc_delay(cor: np.ndarray, nlag: int):
    """Per code period: arg-max lag and the high-resolution-correlator offset of rxcomplex.cpp:630,648-661
    (samples; NaN where the peak is within 2 lags of the window edge, cf. the guard at :635)."""
    pk = cor.argmax(axis=1)
    out = np.full(cor.shape[0], np.nan)
    for p, k in enumerate(pk):
        if k - 2 >= 0 and k + 2 < 2 * nlag + 1:
            c = cor[p]
            narrow = (c[k - 1] - c[k + 1]) / (c[k - 1] - 2.0 * c[k] + c[k + 1])
            wide = (c[k - 2] - c[k + 2]) / (c[k - 2] - 2.0 * c[k] + c[k + 2])
            out[p] = narrow - wide + (k - nlag)
    return pk - nlag, out
