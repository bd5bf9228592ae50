"""Wideband two-station session (BASELINE.json configs[4]): per step, for each station a device-resident 70-Msps int16
capture -> ``twx_fir_decimate_dev`` -> 5-Msps int16 -> the station's two correlations (own code in the loop-back band,
the other station's code in the remote band), every one the full ``processing(d,k)`` of
/root/reference/processing/Octave/godual_ranging.m:11-81 over the step's 1-s windows; the four results are what
/root/reference/acquisition/go_1s.m:88,120,147,171 names oplo, opre, ltlo, ltre.

The session is the stream plumbing around the C ABI, nothing else computes here:

* four contexts (``Correlator``), each with its own HIP streams; a station's front end runs on its loop-back context's
  stream, the remote context waits for it through an event;
* the decimated captures and the result records are DOUBLE-BUFFERED by step parity, so ``submit()`` of step i+1 can be
  enqueued while step i is still running: the front end of step i+1 (bound by vector issue slots) then shares the GPU with
  the correlations of step i (bound by HBM) instead of running in a gap of its own.  The FIR of step i+2 waits (event) for
  the remote correlation of step i to have read the buffer it overwrites;
* ``fetch(step)`` waits for that step's four events only.

There is no CPU path: every call goes to libtwstft_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .correlator import Correlator, band_godual


def godual_plan(stations=("OP", "LTFB"), fs: float = 5e6, n: int = 5_000_000) -> dict:
    """name -> (capture's station, code's station, inclusive band) for the two-station session of godual_ranging.m:83-89:
    ``<st>lo`` = own code around 0 Hz, ``<st>re`` = the other station's code in the +-100 kHz band (sign by station)."""
    a, b = stations
    return {a + "lo": (a, a, band_godual(fs, n)), a + "re": (a, b, band_godual(fs, n, remote=1, OP=0)),
            b + "lo": (b, b, band_godual(fs, n)), b + "re": (b, a, band_godual(fs, n, remote=1, OP=1))}


class WidebandSession:
    """``codes``: station -> chips (uint8, one code period at the narrowband rate ``fs``); ``taps``/``dec``: the front end;
    ``windows``: 1-s windows per step and station; ``plan``: see :func:`godual_plan`."""

    def __init__(self, codes: dict, taps, dec: int, *, fs: float = 5e6, windows: int = 1, device: int = 0, precision: str = "f32",
                 plan: dict | None = None, depth: int = 2, **ctx_kw):
        import torch
        self._torch = torch
        self.lib = L.load()
        self.dev = torch.device("cuda", device)
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.dec, self.W, self.depth = int(dec), int(windows), int(depth)
        if self.depth < 1:
            raise ValueError("depth must be at least 1")
        self.ctx = {}
        try:
            if plan is None:
                sps = ctx_kw.get("sps", 2)
                plan = godual_plan(tuple(codes), fs, len(next(iter(codes.values()))) * sps)
            self.plan = plan
            for k, (_, code_st, _) in plan.items():
                self.ctx[k] = Correlator(codes[code_st], fs=fs, Nint=1, device=device, precision=precision, **ctx_kw)
            # the front end of step i+1 runs beside the correlations of step i BY DESIGN: the matrix-core form of the FIR, which makes
            # packed-fp32 arithmetic of co-resident waves go wrong, is ruled out here whatever TWX_FIR_MFMA says (the library would
            # serialise it against the correlations anyway: csrc/twx_internal.h) — the session always runs the vector form
            for c in self.ctx.values():
                L.check(self.lib.twx_set_option(c._h, L.TWX_OPT_FIR_MFMA, 0), c._h)
            self.N = next(iter(self.ctx.values())).n
            if any(c.n != self.N for c in self.ctx.values()):
                raise ValueError("the codes of a session must have one length")
        except Exception:
            self.close()
            raise
        self.bands = {k: L.twx_band(*v[2]) for k, v in self.plan.items()}
        self.stations = list(dict.fromkeys(v[0] for v in self.plan.values()))
        # the context whose stream runs a station's front end: its first correlation in plan order
        self.front = {st: next(k for k, v in self.plan.items() if v[0] == st) for st in self.stations}
        self.n_out = self.W * self.N
        self.n_in = (self.n_out - 1) * self.dec + int(self.taps.size)
        self.nar = {st: [torch.empty((self.n_out, 2), dtype=torch.int16, device=self.dev) for _ in range(self.depth)] for st in self.stations}
        self.res = {k: [torch.zeros((self.W, C.sizeof(L.twx_result)), dtype=torch.uint8, device=self.dev) for _ in range(self.depth)] for k in self.plan}
        self.stream = {k: torch.cuda.ExternalStream(int(self.lib.twx_stream(c._h)), device=self.dev) for k, c in self.ctx.items()}
        self._fir_done = {st: [None] * self.depth for st in self.stations}
        self._corr_done = {k: [None] * self.depth for k in self.plan}
        self.step = 0

    # -- one step --------------------------------------------------------------------------------------------------
    def submit(self, wide: dict, n_in: int | None = None) -> int:
        """Enqueue one step: ``wide[station]`` = device pointer of that station's int16 ``[I Q]`` capture of ``n_in`` samples
        (default: exactly what ``windows`` seconds need).  Returns the step number for :meth:`fetch`.  Asynchronous."""
        torch = self._torch
        n_in = self.n_in if n_in is None else int(n_in)
        if (n_in - self.taps.size) // self.dec + 1 < self.n_out:
            raise ValueError("capture shorter than %d windows behind the front end" % self.W)
        i, p = self.step, self.step % self.depth
        for st in self.stations:
            f = self.front[st]
            # the buffer this step overwrites was read by every correlation of step i - depth
            for k, v in self.plan.items():
                if v[0] == st and k != f and self._corr_done[k][p] is not None:
                    self.stream[f].wait_event(self._corr_done[k][p])
            self.ctx[f].fir_decimate_dev(int(wide[st]), n_in, self.taps, self.dec, out_i16_dev=self.nar[st][p].data_ptr())
            ev = torch.cuda.Event()
            ev.record(self.stream[f])
            self._fir_done[st][p] = ev
        for k, (st, _, _) in self.plan.items():
            if k != self.front[st]:
                self.stream[k].wait_event(self._fir_done[st][p])
            c = self.ctx[k]
            L.check(self.lib.twx_process_windows_dev(c._h, self.nar[st][p].data_ptr(), self.W, 1, 0, C.byref(self.bands[k]), None,
                                                     self.res[k][p].data_ptr()), c._h)
            ev = torch.cuda.Event()
            ev.record(self.stream[k])
            self._corr_done[k][p] = ev
        self.step += 1
        return i

    def fetch(self, step: int) -> dict:
        """name -> list of ``twx_result`` (one per window) of a submitted step that has not been overwritten
        (``step > self.step - depth - 1``); waits for that step's correlations only."""
        if not (self.step - self.depth <= step < self.step):
            raise ValueError("step %d is not held any more (depth %d)" % (step, self.depth))
        p = step % self.depth
        out = {}
        for k in self.plan:
            self._corr_done[k][p].synchronize()
            raw = self.res[k][p].cpu().numpy().tobytes()
            out[k] = list((L.twx_result * self.W).from_buffer_copy(raw))
        return out

    def decimated(self, station: str, step: int):
        """The step's 5-Msps int16 capture of a station (device tensor; valid until ``depth`` more steps are submitted)."""
        return self.nar[station][step % self.depth]

    def synchronize(self):
        for c in self.ctx.values():
            if c is not None:
                c.synchronize()

    def close(self):
        for c in self.ctx.values():
            if c is not None:
                c.close()
        self.ctx = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
