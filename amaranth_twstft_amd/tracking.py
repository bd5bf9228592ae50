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


_STATE_KEYS = ("fs", "duration", "psbb", "fc", "df", "phi", "last_phi", "pt", "fc_prev", "pt_prev")


def _state_in(state: dict) -> L.twx_track_state:
    st = L.twx_track_state()
    st.fs, st.duration, st.psbb = float(state["fs"]), float(state["duration"]), float(state.get("psbb", 1.0))
    st.fc, st.df, st.phi = float(state["fc"]), float(state.get("df", 0.0)), float(state.get("phi", 0.0))
    st.last_phi, st.pt = float(state["last_phi"]), int(state["pt"])
    st.fc_prev, st.pt_prev = float(state.get("fc_prev", 0.0)), int(state.get("pt_prev", 0))
    return st


def _state_out(st: L.twx_track_state, state: dict, r: L.twx_track_result):
    if not r.updated:
        return None
    for k in ("fc", "df", "phi", "last_phi", "fc_prev"):
        state[k] = float(getattr(st, k))
    state["pt"], state["pt_prev"] = int(st.pt), int(st.pt_prev)
    return dict(freq=r.freq, phi=r.phi, cnt=int(r.cnt), gd=r.gd, dg=r.dg, sdgd=r.sdgd, pk=r.pk)


def tracking_update(cor: np.ndarray, phi: np.ndarray, nlag: int, state: dict, nobs: int | None = None) -> dict | None:
    """One tracking epoch of experiments/231001_DLL_PLL/rxcomplex.cpp:620-745 on the ``bps-1`` code periods whose power/phase
    matrices ``cor``/``phi`` ([bps-1, 2·nlag+1]) came from :func:`sliding_dot` + :func:`get_cor_and_phi` — binding of the
    library's ``twx_track_update`` (host arithmetic in C++, no GPU needed): per-code peak and high-resolution-correlator delay
    (:630-661), 3-sigma filter on median/IQR (:689-700), BPSK half-cycle phase unwrap against ``last_phi`` (:703-716), weighted
    linear fits of phase → carrier update and of delay → code-phase update (:728-745).

    ``state`` holds ``fc pt last_phi fs duration psbb`` (the ``ci[i]`` fields) and is updated in place; the returned dict has
    the printed quantities (``freq phi cnt gd dg sdgd pk``).  ``None`` when no more than half of the periods produced a usable
    peak (:667), in which case ``state`` is left alone.  UNPINNED (GSL/CBLAS program, cannot be built here).

    With ``nobs`` (samples per code period) the result also carries ``mai``: the per-period records the real-sample program
    keeps for its interference cancellation (``twx_track_update_mai``; rx.cpp:664-666,752-757: ``pk_idx amp phase``)."""
    cor = np.ascontiguousarray(cor, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    st, r = _state_in(state), L.twx_track_result()
    if nobs is None:
        L.check(L.load().twx_track_update(cor.ctypes.data_as(C.c_void_p), phi.ctypes.data_as(C.c_void_p), cor.shape[0] + 1, int(nlag),
                                          C.byref(st), C.byref(r)))
        return _state_out(st, state, r)
    bps = cor.shape[0] + 1
    pk, amp, ph = np.zeros(bps, dtype=np.int32), np.zeros(bps), np.zeros(bps)
    mai = L.twx_track_mai(pk.ctypes.data_as(C.POINTER(C.c_int32)), amp.ctypes.data_as(C.POINTER(C.c_double)), ph.ctypes.data_as(C.POINTER(C.c_double)))
    L.check(L.load().twx_track_update_mai(cor.ctypes.data_as(C.c_void_p), phi.ctypes.data_as(C.c_void_p), bps, int(nlag), int(nobs),
                                          C.byref(st), C.byref(r), C.byref(mai)))
    out = _state_out(st, state, r)
    if out is not None:
        out["mai"] = dict(pk_idx=pk, amp=amp, phase=ph)
    return out


def track_epoch_dev(cor_ctx, iq_dev: int, n_samples: int, replica_dev: int, nobs: int, bps: int, nlag: int, state: dict,
                    scale: float = 1.0, n_channels: int = 1, channel: int = 0) -> dict | None:
    """The whole epoch (rxcomplex.cpp:593-745) on a device-resident capture in one library call (``twx_track_epoch_dev``):
    down-conversion at ``state['fc']`` from ``state['pt']`` with phase ``fmod(pt*fc/fs, 1)`` (:594), ±nlag sliding dot products
    of ``bps-1`` code periods, power/phase, then :func:`tracking_update`'s arithmetic.  ``cor_ctx``: any ``Correlator`` (its
    stream and scratch buffers are used)."""
    st, r = _state_in(state), L.twx_track_result()
    L.check(cor_ctx._lib.twx_track_epoch_dev(cor_ctx._h, iq_dev, int(n_samples), n_channels, channel, int(nobs), int(bps), int(nlag),
                                             replica_dev, float(scale), C.byref(st), C.byref(r)), cor_ctx._h)
    return _state_out(st, state, r)


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
