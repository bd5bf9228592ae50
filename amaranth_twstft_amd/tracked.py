"""Tracked multi-code ranging: the window loop of
acquisition/claudio_aligned_code_ranging_separate.m:143-205 on the device.

The script reads ``ls``-second chunks of a single-channel capture, finds the carrier once
(``search_df`` :27-47), then walks the chunk one code period (40 ms) at a time; whenever the
correlation peak has left the first 43 samples it moves the window start so the peak sits at
sample 21 (:170-180) and measures again.  Here the chunk lives in device memory and the codes of a
chunk are correlated in one batched launch on the assumption that the window did not move; the
results are then scanned in order and the batch is cut at the first code that asks for a
re-alignment ("coarse-parallel + serial fix-up").  In steady state there is one
batch per chunk.

All arithmetic on the samples runs in the HIP library (``twx_process_windows_dev``,
``twx_sqspec_band_dev``, ``twx_sqspec_bins_dev``, ``twx_xcorr_map``); this module is the control
flow only and keeps the script's 1-based bookkeeping and its quirks (see the oracle's
``ranging_tracked`` for the list).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib as L
from .correlator import Correlator


def _oround(x: float) -> int:
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


class _DevBuf:
    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, nbytes
        self.ptr = lib.twx_dev_alloc(nbytes)
        if not self.ptr:
            raise MemoryError("twx_dev_alloc failed")

    def upload(self, offset, arr):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        L.check(self.lib.twx_memcpy_h2d(self.ptr + offset, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, offset, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        L.check(self.lib.twx_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr + offset, nbytes))
        return out

    def close(self):
        if self.ptr:
            self.lib.twx_dev_free(self.ptr)
            self.ptr = None


class TrackedRanging:
    """``processing(d,df)`` + ``search_df`` + the tracked loop, one code per 40-ms block."""

    def __init__(self, chips, fs=5e6, sps=2, Nint=1, ls_samples=None, band_hz=8000.0, df_threshold=20.0,
                 device=-1, precision="f32", max_batch=0):
        self.cor = Correlator(chips, fs=fs, sps=sps, Nint=Nint, convention="claudio", var_ddof=1, device=device,
                              precision=precision, max_batch=max_batch)
        self.fs, self.Nint, self.n = fs, Nint, self.cor.n
        self.L = int(ls_samples if ls_samples is not None else fs * 2)          # fs*ls, ls = 2 (:15,157)
        if self.L % self.n:
            raise ValueError("chunk length must be a whole number of code periods")
        self.freq = np.linspace(-fs / 2, fs / 2 - 1.0, self.L)                   # :131
        self.k = np.nonzero((self.freq < band_hz) & (self.freq > -band_hz))[0]   # :134 (ranging band)
        self.df_threshold = df_threshold
        self._lib = self.cor._lib
        self._buf = _DevBuf(self._lib, (self.L + self.n + 64) * 4)

    def close(self):
        self._buf.close()
        self.cor.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- search_df (:27-47) ----------------------------------------------------------------
    def search_df(self, chunk_i16: np.ndarray) -> int:
        """0-based index into the shifted axis of the accepted carrier bin, -1 if none."""
        chunk = np.ascontiguousarray(chunk_i16, dtype=np.int16).reshape(-1)
        assert chunk.size == 2 * self.L
        self._buf.upload(0, chunk)
        k = self.k
        half = self.L // 2
        d2k = self.cor.sqspec_band_dev(self._buf.ptr, self.L, int(k[0]) - half, k.size)     # d2(k)
        ktmp = np.nonzero(d2k > np.median(d2k) * self.df_threshold)[0] + int(k[0])
        kbon = -1
        if 0 < len(ktmp) < 100:
            r = 2 * self.Nint + 1
            first = chunk[:2 * self.n]
            for kk in ktmp:
                dftmp = self.freq[kk] / 2
                prnmap = np.abs(self.cor.xcorr_map(first, dftmp, raw_mean=True))[::r] * r       # N-point map (:36)
                b = int(prnmap.argmax())
                prnsig = prnmap[b]
                prnmap[max(b - 5, 0):b + 6] = 0
                if prnsig ** 2 / np.var(prnmap, ddof=1) > 100:
                    kbon = int(kk)
        return kbon

    # -- the loop (:143-205) ---------------------------------------------------------------
    def run(self, raw, skip_samples: int = 0, kbon: int | None = None) -> dict:
        raw = np.asarray(raw).reshape(-1)
        n, r, Lc = self.n, 2 * self.Nint + 1, self.L
        out = dict(xval=[], indice1=[], correction1=[], SNR1r=[], SNR1i=[], puissance1=[], df=[], moved=[], movedval=[],
                   kbon=-1, batches=0)
        df_found = kbon is not None
        if df_found:
            out["kbon"] = int(kbon)
        pos = skip_samples * 2
        carry = 0                                   # samples of dold kept at the head of the device buffer
        p = 1
        guard = 0
        while True:
            chunk = raw[pos:pos + 2 * Lc]
            pos += 2 * Lc
            if chunk.size != 2 * Lc:
                break
            if not df_found:
                kb = self.search_df(chunk)
                if kb >= 0:
                    df_found, out["kbon"] = True, kb
                chunk = raw[0:2 * Lc]               # the script re-opens the file here (:153-155)
                pos = 2 * Lc
                guard += 1
                if not df_found and guard > 2:
                    break
            if not df_found:
                continue
            kb = out["kbon"]
            self._buf.upload(carry * 4, np.ascontiguousarray(chunk, dtype=np.int16))
            Ld = carry + Lc
            bins = np.arange(kb - 3, kb + 4) - Ld // 2          # shifted index i ↔ bin i - floor(Ld/2)
            d2 = np.abs(self.cor.sqspec_bins_dev(self._buf.ptr, Ld, bins))
            df = self.freq[int(np.argmax(d2)) + kb - 3] / 2
            out["df"].append(df)
            dindex = 1.0
            done = False
            while not done:
                s0 = _oround(dindex)
                J = 1
                while dindex + J * n + n - 1 <= Ld:
                    J += 1
                res = self.cor.process_dev(self._buf.ptr + (s0 - 1) * 4, J, 1, 0, df=df)
                out["batches"] += 1
                for j, g in enumerate(res):
                    ind = (g.indice + 1) / r
                    o = g
                    stop = False
                    snr = g.SNRi + g.SNRr
                    if snr > 0 and 10 * math.log10(snr) > -30 and ((43 < ind < n / 2) or (n / 2 < ind < n - 2)):
                        out["moved"].append(p)
                        out["movedval"].append(ind + 1)
                        dcur = dindex + j * n
                        dnew = dcur + n if dcur - ind + 1 < 0 else dcur
                        dnew = dnew - ind + 21
                        s1 = _oround(dnew)
                        if s1 >= 1 and s1 - 1 + n <= Ld:
                            o = self.cor.process_dev(self._buf.ptr + (s1 - 1) * 4, 1, 1, 0, df=df)[0]
                            ind = float(o.indice + 1)
                            self._append(out, o, ind)
                            p += 1
                            dindex = dnew + n
                            done = dindex + n - 1 > Ld
                            break                                    # later codes of this batch are stale
                        stop = True
                    self._append(out, o, ind)
                    p += 1
                    if stop:
                        dindex = dindex + (j + 1) * n
                        done = True
                        break
                else:
                    dindex += J * n
                    done = True
            # dold = d(round(dindex):end)  (:196-199): slide the tail to the head of the buffer
            if dindex < Ld:
                s = _oround(dindex) - 1
                tail = self._buf.download(s * 4, (Ld - s) * 4)
                self._buf.upload(0, tail)
                carry = Ld - s
            else:
                carry = 0
        return out

    @staticmethod
    def _append(out, g, ind):
        out["xval"].append(g.xval); out["indice1"].append(ind); out["correction1"].append(g.correction)
        out["SNR1r"].append(g.SNRr); out["SNR1i"].append(g.SNRi); out["puissance1"].append(g.puissance)
        # the script receives these two without an index (:168): its workspace keeps the last code's values
        out["puissancecode"] = getattr(g, "puissancecode", float("nan"))
        out["puissancenoise"] = getattr(g, "puissancenoise", float("nan"))

    def run_file(self, path: str, skip_seconds: float = 30.0, **kw) -> dict:
        """Whole single-channel sc16 capture file (``fseek(f,30*fs*2*2)`` :128 → ``skip_seconds``)."""
        raw = np.memmap(path, dtype=np.int16, mode="r") if os.path.getsize(path) else np.zeros(0, np.int16)
        return self.run(raw, skip_samples=int(skip_seconds * self.fs), **kw)
