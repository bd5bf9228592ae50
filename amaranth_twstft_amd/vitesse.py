"""Host-side mirror of experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m on the HIP library: the two-channel capture loop in which the
returned signal (channel 1) is correlated after a velocity-compensating resampling — ``yi=interp1([0:N-1],y,[0:N-1]*1/(1-vitesse)+t0)``
with ``t0`` carried from window to window (:40-43, :68-71) — and the reference (channel 2) plainly (:47), both against the zero-mean
0/1 replica (:7-10) without interpolation.  Nothing is computed here but the script's last lines (``solution12 - solution22``): the
per-window work is ``twx_process_file`` on a context with ``twx_set_resample``.
"""
from __future__ import annotations

import numpy as np

from .correlator import Correlator, freq_axis


def band_vitesse(fs: float, n: int, lo: float = 96200.0, hi: float = 106200.0) -> tuple[int, int]:
    """``k=find((freq<106200)&(freq>96200))`` (:32) as an inclusive 0-based index range."""
    f = freq_axis(fs, n)
    k = np.nonzero((f < hi) & (f > lo))[0]
    return int(k[0]), int(k[-1])


def ranging_vitesse(path: str, chips, fs: float = 5e6, vitesse: float = -3.25e-9, band_hz=(96200.0, 106200.0), device: int = -1,
                    max_windows: int | None = None, precision: str = "f32") -> dict:
    """The loop of :17-74 over a two-channel int16 capture ``[I1 Q1 I2 Q2]``.  Returns the script's vectors: ``indice1`` (with ``dt``
    added, :68), ``indice2`` (1-based, as Octave prints them, :50), ``correction12`` / ``correction22``, ``df``, ``xval1`` / ``xval2``,
    ``solution12`` / ``solution22`` and ``delay = (solution12 - solution22) / fs`` (:76-80)."""
    n = 2 * len(chips)
    kw = dict(fs=fs, Nint=0, code_levels="unipolar", code_zero_mean=True, device=device, precision=precision)
    with Correlator(chips, **kw) as c1, Correlator(chips, **kw) as c2:
        c1.set_resample(vitesse)
        r1 = c1.process_file(path, n_channels=2, channel=0, band=band_vitesse(fs, n, *band_hz), max_windows=max_windows)
        r2 = c2.process_file(path, n_channels=2, channel=1, df=0.0, max_windows=max_windows)            # d2 is correlated unmixed (:47)
    indice1 = np.array([r.indice + 1 + r.dt for r in r1], dtype=float)
    indice2 = np.array([r.indice + 1 for r in r2], dtype=float)
    c12 = np.array([r.correction for r in r1])
    c22 = np.array([r.correction for r in r2])
    s12, s22 = indice1 + c12, indice2 + c22
    return dict(indice1=indice1, indice2=indice2, correction12=c12, correction22=c22, df=np.array([r.df for r in r1]),
                xval1=np.array([r.xval for r in r1]), xval2=np.array([r.xval for r in r2]), status=np.array([r.status for r in r1]),
                solution12=s12, solution22=s22, delay=(s12 - s22) / fs)
