"""CPU: control flow of amaranth_twstft_amd/tracked.py (speculative batches + serial re-alignment fix-up) with the
device calls replaced by the oracle — the same loop must reproduce oracle.ranging_tracked exactly."""
import types

import numpy as np

from amaranth_twstft_amd import synth, tracked
from oracle import twstft_oracle as orc
from tests.helpers import chips_for

FS = 5e6


class _FakeBuf:
    def __init__(self, nbytes):
        self.mem = np.zeros(nbytes, dtype=np.uint8)
        self.ptr = 1 << 20                       # pretend device address

    def upload(self, offset, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        self.mem[offset:offset + b.size] = b

    def download(self, offset, nbytes):
        return self.mem[offset:offset + nbytes].copy()

    def close(self):
        pass


class _FakeCor:
    """process_dev / sqspec_* / xcorr_map computed by the oracle on the fake buffer's contents."""

    def __init__(self, chips, buf, n):
        self.buf, self.n, self.calls = buf, n, 0
        self.code = orc.make_code(chips, 2)
        self.fc = orc.make_fcode(self.code, "claudio")
        self.temps = np.arange(n) / FS

    def _samples(self, ptr, count):
        off = ptr - self.buf.ptr
        raw = self.buf.mem[off:off + count * 4].view(np.int16)
        return raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)

    def process_dev(self, ptr, nwin, nch, ch, band=None, df=None):
        self.calls += 1
        out = []
        for w in range(nwin):
            d = self._samples(ptr + w * self.n * 4, self.n)
            o = orc.processing_claudio(d - d.mean(), df, self.temps, self.fc, self.code, Nint=1, ddof=1)
            out.append(types.SimpleNamespace(indice=o["indice"], correction=o["correction"], xval=o["xval"], SNRr=o["SNRr"],
                                             SNRi=o["SNRi"], puissance=o["puissance"]))
        return out

    def sqspec_bins_dev(self, ptr, L, bins):
        d = self._samples(ptr, L)
        f = np.fft.fft(d ** 2)
        return f[np.asarray(bins) % L]

    def sqspec_band_dev(self, ptr, L, k_lo, nk):
        d = self._samples(ptr, L)
        f = np.abs(np.fft.fft(d ** 2))
        return f[np.arange(k_lo, k_lo + nk) % L]

    def xcorr_map(self, first, df, raw_mean=False):
        d = first[0::2].astype(np.float64) + 1j * first[1::2].astype(np.float64)
        y = d * np.exp(-2j * np.pi * df * self.temps)
        m = np.fft.ifft(self.fc * np.conj(np.fft.fft(y)))
        return np.repeat(m, 3) / 3.0              # every 3rd sample is what search_df looks at

    def close(self):
        pass


def _make(chips, n, Lc):
    tr = object.__new__(tracked.TrackedRanging)
    tr.fs, tr.Nint, tr.n, tr.L = FS, 1, n, Lc
    tr.freq = np.linspace(-FS / 2, FS / 2 - 1.0, Lc)
    tr.k = np.nonzero((tr.freq < 8000.0) & (tr.freq > -8000.0))[0]
    tr.df_threshold = 20.0
    tr._buf = _FakeBuf((Lc + n + 64) * 4)
    tr.cor = _FakeCor(chips, tr._buf, n)
    tr._lib = None
    return tr


def _capture(ncodes, delay, seed):
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(30.0, FS), phi0=5, amp=500,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=seed)
    return chips, n, synth.synth_channel(n * ncodes, chips, 2, p)


def test_tracked_control_flow_matches_the_oracle_loop():
    chips, n, a = _capture(45, 1500, 4)
    _, _, b = _capture(45, 1500 + 777, 5)                 # a delay jump in the middle forces a second re-alignment
    raw = np.concatenate((a, b))
    Lc = 30 * n
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc)
    tr = _make(chips, n, Lc)
    got = tr.run(raw)
    assert got["kbon"] == want["kbon"] > 0 and got["df"] == want["df"]
    assert got["moved"] == want["moved"] and len(want["moved"]) >= 2
    assert got["indice1"] == want["indice1"] and got["movedval"] == want["movedval"]
    assert np.allclose(got["correction1"], want["correction1"]) and np.allclose(got["xval"], want["xval"])
    # speculation: one batch per chunk plus one per re-alignment (batch cut + 1-window re-measure)
    assert got["batches"] <= len(want["df"]) + 2 * len(want["moved"])
