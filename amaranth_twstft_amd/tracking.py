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


def _wlinear(x, w, y):
    """``gsl_fit_wlinear``: weighted least squares y = c0 + c1·x; returns (c0, c1, chisq)."""
    x, w, y = (np.asarray(v, dtype=float) for v in (x, w, y))
    W = w.sum()
    xm, ym = (w * x).sum() / W, (w * y).sum() / W
    dx, dy = x - xm, y - ym
    c1 = (w * dx * dy).sum() / (w * dx * dx).sum()
    c0 = ym - c1 * xm
    return c0, c1, float((w * (y - c0 - c1 * x) ** 2).sum())


def tracking_update(cor: np.ndarray, phi: np.ndarray, nlag: int, state: dict) -> dict | None:
    """One tracking epoch of experiments/231001_DLL_PLL/rxcomplex.cpp:620-745 on the ``bps-1`` code periods
    whose power/phase matrices ``cor``/``phi`` ([bps-1, 2·nlag+1]) came from :func:`sliding_dot` +
    :func:`get_cor_and_phi`: per-code peak and high-resolution-correlator delay (:630-661), 3-sigma filter on
    median/IQR (:689-716), BPSK half-cycle phase unwrap against ``last_phi`` (:710-715), weighted linear fits of
    phase → carrier update and of delay → code-phase update (:728-745).

    ``state`` holds ``fc pt last_phi fs duration psbb`` (the ``ci[i]`` fields) and is updated in place; the
    returned dict has the printed quantities (``freq phi cnt gd dg sdgd pk``).  ``None`` when fewer than half of
    the periods produced a usable peak (:667), in which case ``state`` is left alone.  UNPINNED (GSL/CBLAS
    program, cannot be built here).
    """
    bps = cor.shape[0] + 1
    fs, duration, pt = state["fs"], state["duration"], state["pt"]
    psbb = state.get("psbb", 1.0)
    res_gd, res_phi, ps, w = (np.zeros(bps) for _ in range(4))
    ttag_phi = np.zeros(bps)
    nl = 2 * nlag + 1
    for p in range(bps - 1):
        k = int(np.argmax(cor[p]))                                    # cblas_idamax on non-negative powers (:630)
        ttag_phi[p] = p * duration + pt / fs
        ps[p] = cor[p, k] / psbb
        if k - 2 >= 0 and k + 2 < nl:
            c = cor[p]
            res_phi[p] = phi[p, k]
            res_gd[p] = ((c[k - 1] - c[k + 1]) / (c[k - 1] - 2.0 * c[k] + c[k + 1])
                         - (c[k - 2] - c[k + 2]) / (c[k - 2] - 2.0 * c[k] + c[k + 2])
                         + float(pt + k - nlag)) * 1.0e9 / fs
            w[p] = 1.0
    cnt = int(w.sum())
    if cnt * 2 <= bps:
        return None
    sel = np.sort(res_gd[w > 0])                                      # kth_smallest ≡ order statistics
    ii = sel.size
    c0 = sel[ii // 2]
    stddev = (sel[ii * 3 // 4] - sel[ii // 4]) / 1.349
    last_phi = state["last_phi"]
    cnt = 0
    for p in range(bps - 1):
        if w[p] != 0.0:
            if abs(res_gd[p] - c0) < 3.0 * stddev:
                cnt += 1
                while abs(res_phi[p] - last_phi) > 0.25:
                    res_phi[p] += -0.5 if res_phi[p] > last_phi else 0.5
                last_phi = res_phi[p]
            else:
                w[p] = 0.0
    state["last_phi"] = last_phi
    c0, c1, _ = _wlinear(ttag_phi, w, res_phi)
    state["fc_prev"] = state["fc"]
    state["fc"] += round(c1)
    state["df"] = c1 - round(c1)
    state["phi"] = float(np.fmod(c0 + 1000.0, 1.0))
    ttag_gd = np.arange(bps) * duration
    g0, g1, chi = _wlinear(ttag_gd, w, res_gd)
    out = dict(freq=state["fc"] + state["df"], phi=state["phi"], cnt=cnt, gd=g0 + 0.5 * g1, dg=g1,
               sdgd=float(np.sqrt(chi / cnt)), pk=float(ps[w > 0].mean()) if cnt else 0.0)
    state["pt_prev"] = pt
    state["pt"] = int(round((g0 + g1) * fs / 1.0e9))
    return out


def prn_sampling(nobs: int, code, rc: float, fs: float, delay_ns: float = 0.0) -> np.ndarray:
    """Replica sampled at ``fs`` from a chip sequence clocked at ``rc`` chips/s and delayed by ``delay_ns``:
    ``idx=floor(fmod((i/fs-delay*1e-9)*rc, clen))`` wrapped into [0, clen) — ``PRN_sampling`` of
    experiments/231001_DLL_PLL/rxcomplex.cpp:965-978 (any fs/rc ratio, fractional delays).  ``code`` holds the chip
    VALUES (±1 as ``SDRcode`` produces them); returns float32 of length ``nobs`` for :func:`sliding_dot`."""
    code = np.asarray(code)
    clen = code.size
    i = np.arange(nobs, dtype=np.float64)
    idx = np.floor(np.fmod((i / fs - delay_ns * 1.0e-9) * rc, float(clen))).astype(np.int64)
    idx = np.where(idx < 0, idx + clen, idx)
    idx = np.where(idx >= clen, idx - clen, idx)
    return code[idx].astype(np.float32)
