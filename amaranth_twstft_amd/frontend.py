"""FIR decimating front end for wideband captures (BASELINE.json configs[4]: 70 Msps → 5 Msps).

The reference has no 70 Msps capture and no decimating front end for this path; the only FIRs in the
repository are GNU Radio ``firdes.low_pass`` blocks (experiments/2403/zmq_rx.py:208-215,
experiments/2403/x310.grc:251-290).  ``lowpass_taps`` follows that design rule (Hamming windowed
sinc, unit DC gain); the filtering itself runs in ``twx_fir_decimate`` on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def lowpass_taps(fs: float, cutoff: float, transition: float) -> np.ndarray:
    """``firdes.low_pass(1, fs, cutoff, transition, WIN_HAMMING)``: odd ntaps = 53*fs/(22*transition)."""
    ntaps = int(53.0 * fs / (22.0 * transition))
    if ntaps % 2 == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    n = np.arange(-m, m + 1)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * (n + m) / (ntaps - 1))
    fw = 2 * np.pi * cutoff / fs
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(n == 0, fw / np.pi, np.sin(n * fw) / (n * np.pi)) * w
    return (h / h.sum()).astype(np.float32)


def fir_decimate(raw, taps, dec: int, n_channels: int = 1, channel: int = 0, out: str = "int16"):
    """y[m] = sum_j taps[j]*x[m*dec+j] on the GPU.  ``out``: "int16" → int16 [nout,2]; "f32" → complex64."""
    lib = L.load()
    raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1)
    n_in = raw.size // (2 * n_channels)
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    nout_max = (n_in - taps.size) // dec + 1
    if nout_max < 1:
        raise ValueError("capture shorter than the filter")
    y16 = np.empty((nout_max, 2), dtype=np.int16) if out == "int16" else None
    yf = np.empty(nout_max, dtype=np.complex64) if out == "f32" else None
    nout = C.c_int64()
    L.check(lib.twx_fir_decimate(raw.ctypes.data_as(C.c_void_p), n_in, n_channels, channel, taps.ctypes.data_as(C.c_void_p),
                                 taps.size, dec, y16.ctypes.data_as(C.c_void_p) if y16 is not None else None,
                                 yf.ctypes.data_as(C.c_void_p) if yf is not None else None, C.byref(nout)))
    return (y16 if y16 is not None else yf)[: nout.value]
