"""Deterministic synthetic PRN+noise captures (integer-only, bit-reproducible).

The reference has no capture fixtures for the correlation path, so tests and the
benchmark run on synthetic captures in the reference's wire format: little-endian int16,
``[I Q]`` per sample (X310-era single-channel files, acquisition/rx_multi_samples.cpp:155,214-215)
or ``[I1 Q1 I2 Q2]`` (B210-era two-channel files, processing/Octave/godual_ranging.m:76-79).

Everything here is integer arithmetic (64-bit), so the numpy implementation below and the HIP
generator ``twx_synth_capture`` (csrc/twx_synth.hip) produce identical bytes on any machine:

* chip index    c(n) = floor(((n*256 - delay_q8) mod (L*sps*256)) / (sps*256)),  value 2*chip-1
* carrier       32-bit phase accumulator ph(n) = phi0 + n*fstep (mod 2^32); cos/sin from a
                fixed-point odd/even polynomial in the first quadrant (Q30)
* noise         Irwin-Hall(8) of 16-bit uniforms from a splitmix64 counter hash, per I and Q
* sample        clip(((amp*c*cos) >> 30) + ((noise_sum*noise_gain) >> 20), int16)

The realised carrier offset is exactly ``fstep * fs / 2^32`` Hz.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M64 = (1 << 64) - 1
_PI2_Q30 = 1686629713  # round(pi/2 * 2^30)
_IH8_STD = 53510.53    # std of the sum of 8 uniform 16-bit integers: sqrt(8*(65536^2-1)/12)

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    x = x.copy()
    x ^= x >> np.uint64(30)
    x *= _C1
    x ^= x >> np.uint64(27)
    x *= _C2
    x ^= x >> np.uint64(31)
    return x


def _icossin_q30(ph: np.ndarray):
    """cos/sin (Q30, int64) of a uint32 phase (full turn = 2^32). Pure integer ops."""
    ph = ph.astype(np.int64)
    quad = ph >> 30
    x = ph & ((1 << 30) - 1)              # fraction of the quadrant, Q30
    y = (x * _PI2_Q30) >> 30              # angle in radians, Q30, 0..pi/2

    def poly(y):
        y2 = (y * y) >> 30
        # sin: y*(1 - y2/6*(1 - y2/20*(1 - y2/42*(1 - y2/72*(1 - y2/110)))))
        one = np.int64(1 << 30)
        t = one - y2 // 110
        t = one - ((y2 // 72) * t >> 30)
        t = one - ((y2 // 42) * t >> 30)
        t = one - ((y2 // 20) * t >> 30)
        t = one - ((y2 // 6) * t >> 30)
        s = (y * t) >> 30
        # cos: 1 - y2/2*(1 - y2/12*(1 - y2/30*(1 - y2/56*(1 - y2/90*(1 - y2/132)))))
        u = one - y2 // 132
        u = one - ((y2 // 90) * u >> 30)
        u = one - ((y2 // 56) * u >> 30)
        u = one - ((y2 // 30) * u >> 30)
        u = one - ((y2 // 12) * u >> 30)
        c = one - ((y2 // 2) * u >> 30)
        return c, s

    c0, s0 = poly(y)
    cs = np.where(quad == 0, c0, np.where(quad == 1, -s0, np.where(quad == 2, -c0, s0)))
    sn = np.where(quad == 0, s0, np.where(quad == 1, c0, np.where(quad == 2, -s0, -c0)))
    return cs, sn


@dataclass
class SynthParams:
    """Parameters of one synthetic channel-window (all integers)."""
    delay_q8: int          # code delay in 1/256 sample
    fstep: int             # carrier step per sample, units of 2^-32 turn (two's complement OK)
    phi0: int = 0          # initial carrier phase, 2^-32 turn
    amp: int = 200         # signal amplitude (LSB)
    noise_gain: int = 0    # see noise_gain_for_sigma()
    seed: int = 1
    stream: int = 0        # distinguishes windows/channels under one seed


def fstep_for_df(df_hz: float, fs: float) -> int:
    """Nearest accumulator step for ``df_hz``; the realised offset is ``df_of_fstep``."""
    return int(round(df_hz / fs * 2.0 ** 32)) & 0xFFFFFFFF


def df_of_fstep(fstep: int, fs: float) -> float:
    s = fstep & 0xFFFFFFFF
    if s >= 1 << 31:
        s -= 1 << 32
    return s * fs / 2.0 ** 32


def noise_gain_for_sigma(sigma: float) -> int:
    """Integer gain giving per-component noise std ≈ ``sigma`` LSB."""
    return int(round(sigma * (1 << 20) / _IH8_STD))


def synth_channel(n: int, chips: np.ndarray, sps: int, p: SynthParams, n0: int = 0) -> np.ndarray:
    """int16 array [n, 2] (I, Q) for sample indices n0 .. n0+n-1 of one channel."""
    chips = np.asarray(chips, dtype=np.uint8)
    L = int(chips.size)
    idx = np.arange(n0, n0 + n, dtype=np.int64)
    period = L * sps * 256
    cidx = ((idx * 256 - int(p.delay_q8)) % period) // (sps * 256)
    c = chips[cidx].astype(np.int64) * 2 - 1
    ph = ((idx.astype(np.uint64) * np.uint64(p.fstep & 0xFFFFFFFF)) + np.uint64(p.phi0 & 0xFFFFFFFF)) \
        & np.uint64(0xFFFFFFFF)
    cs, sn = _icossin_q30(ph)
    amp = np.int64(p.amp)
    si = (amp * c * cs) >> 30
    sq = (amp * c * sn) >> 30
    if p.noise_gain:
        key = np.uint64(((p.seed & _M64) * 0xD6E8FEB86659FD93 + (p.stream & _M64) * 0xA0761D6478BD642F) & _M64)
        ctr = idx.astype(np.uint64) * _GOLD + key
        h1 = _mix64(ctr)
        h2 = _mix64(ctr ^ np.uint64(0x5851F42D4C957F2D))
        m16 = np.uint64(0xFFFF)

        # I uses the low halves of both hashes, Q the high halves → 8 uniforms each
        def parts(h):
            lo = ((h & m16) + ((h >> np.uint64(16)) & m16)).astype(np.int64)
            hi = (((h >> np.uint64(32)) & m16) + (h >> np.uint64(48))).astype(np.int64)
            return lo, hi
        a_lo, a_hi = parts(h1)
        b_lo, b_hi = parts(h2)
        h3 = _mix64(ctr ^ np.uint64(0x2545F4914F6CDD1D))
        h4 = _mix64(ctr ^ np.uint64(0x9FB21C651E98DF25))
        c_lo, c_hi = parts(h3)
        d_lo, d_hi = parts(h4)
        ni = a_lo + b_lo + c_lo + d_lo - 4 * 65535
        nq = a_hi + b_hi + c_hi + d_hi - 4 * 65535
        g = np.int64(p.noise_gain)
        si = si + ((ni * g) >> 20)
        sq = sq + ((nq * g) >> 20)
    out = np.empty((n, 2), dtype=np.int16)
    out[:, 0] = np.clip(si, -32768, 32767)
    out[:, 1] = np.clip(sq, -32768, 32767)
    return out


def synth_capture(n: int, chips: np.ndarray, sps: int, channels: list[SynthParams], n0: int = 0) -> np.ndarray:
    """Interleaved capture: int16 [n, 2*len(channels)] = ``[I1 Q1 I2 Q2 ...]`` per sample."""
    cols = [synth_channel(n, chips, sps, p, n0) for p in channels]
    return np.concatenate(cols, axis=1)
