"""The DLL/PLL receiver of experiments/231001_DLL_PLL/rxcomplex.cpp as a drop-in: ``sdr.param`` in, ``ch?.pn??.????kcps.dat``
rows out — binding of ``twx_rx_*`` (include/twstft_hip.h).  Everything (parameter parsing, replica set-up on the device,
x2 interpolation, acquisition-or-tracking per channel and second, the rows and the log lines) is in libtwstft_hip.so; this
module marshals arguments and is the ``./rxcomplex data.bin sdr.param`` command line:

    python -m amaranth_twstft_amd.receiver data.bin sdr.param [--codes DIR] [--out DIR] [--seed N] [--seconds N] [--real]

``--real`` / ``Receiver(..., real=True)`` is the other program of that directory, ``rx.cpp``: the I samples only, no
interpolation, ``S`` rows of the parameter file with successive interference cancellation (rx.cpp:505-518), ``rxreal.log``.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np

from . import _lib as L

STATUS = {L.TWX_RX_NO_SIGNAL: "no signal", L.TWX_RX_ACQUIRED: "acquired", L.TWX_RX_CODE_LOCK: "code lock", L.TWX_RX_TRACKED: "tracked",
          L.TWX_RX_ACQ_FAILED: "acq failed", L.TWX_RX_LOCK_LOST: "lock lost"}


def parse_param(path: str, max_rows: int = 120):
    """Rows of a parameter file by the program's rules (rxcomplex.cpp:263-296)."""
    lib = L.load()
    rows = (L.twx_rx_row * max_rows)()
    n = lib.twx_rx_parse_param(os.fsencode(path), rows, max_rows)
    if n < 0:
        raise L.TwxError(n, (lib.twx_rx_last_error(None) or b"?").decode())
    return [rows[i] for i in range(n)]


def make_row(ch: str, pn: int, fc_init: float, frange: float, fstep: float, snr_min_db: float, kcps: int = 2500, fltkhz: float = 1250.0,
             mode: str = "N", code=None) -> L.twx_rx_row:
    r = L.twx_rx_row()
    r.ch, r.mode, r.pn, r.fc_init, r.kcps, r.fltkhz = ch.encode(), mode.encode(), pn, fc_init, kcps, fltkhz
    r.frange, r.fstep, r.snr_min_db = frange, fstep, snr_min_db
    if code is not None:
        arr = np.ascontiguousarray(code, dtype=np.uint8)
        r._keep = arr                                         # the library copies the chips during twx_rx_create
        r.code, r.code_len = arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.size
    return r


class Receiver:
    def __init__(self, rows, fs_in: float = 5e6, code_dir: str | None = None, out_dir: str | None = None, seed: int = 1, acq_block: int = -1,
                 dec_a: int = 1, device: int = -1, real: bool = False):
        self._lib = L.load()
        cfg = L.twx_rx_config()
        ninterp = 1 if real else 2
        cfg.fs_in, cfg.ninterp, cfg.dec_a = fs_in, ninterp, dec_a
        cfg.code_dir = os.fsencode(code_dir) if code_dir else None
        cfg.out_dir = os.fsencode(out_dir) if out_dir else None
        cfg.seed, cfg.acq_block, cfg.device = seed, acq_block, device
        arr = (L.twx_rx_row * len(rows))(*rows)
        h = C.c_void_p()
        rc = self._lib.twx_rx_create(C.byref(cfg), arr, len(rows), C.byref(h))
        if rc == L.TWX_E_SIZE:
            # transform lengths without a plan pair yet (nobs, nfft, the capture second): build the plug-ins, retry
            from . import plans
            n_in = int(round(fs_in))
            for r in rows:
                bps = 2500000 // (10000 if r.pn < 100 else 100000)
                nobs = ninterp * n_in // bps
                nfft = 1
                while True:
                    nfft *= 2
                    if nfft > nobs * 2 // dec_a:
                        break
                for n, prec in ((nobs, 1), (nfft, 1), (nfft, 0)):
                    plans.ensure(n, prec, self._lib)
            if not real:
                plans.ensure(n_in, 0, self._lib)
            rc = self._lib.twx_rx_create(C.byref(cfg), arr, len(rows), C.byref(h))
        if rc:
            raise L.TwxError(rc, (self._lib.twx_rx_last_error(None) or b"?").decode())
        self._h, self.n_rows, self.n_in = h, len(rows), int(round(fs_in))

    def _check(self, rc):
        if rc:
            raise L.TwxError(rc, (self._lib.twx_rx_last_error(self._h) or b"?").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.twx_rx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def channel(self, i: int) -> L.twx_rx_channel_info:
        info = L.twx_rx_channel_info()
        self._check(self._lib.twx_rx_channel(self._h, i, C.byref(info)))
        return info

    def second(self, raw):
        """One second of ``[IA QA IB QB]`` int16 frames (host array) -> one report per parameter row."""
        raw = np.ascontiguousarray(raw, dtype=np.int16).reshape(-1)
        if raw.size != 4 * self.n_in:
            raise ValueError("one second is %d int16 values" % (4 * self.n_in))
        rep = (L.twx_rx_report * self.n_rows)()
        self._check(self._lib.twx_rx_second(self._h, raw.ctypes.data_as(C.c_void_p), rep))
        return list(rep)

    def second_dev(self, iq_dev: int):
        rep = (L.twx_rx_report * self.n_rows)()
        self._check(self._lib.twx_rx_second_dev(self._h, iq_dev, rep))
        return list(rep)

    def run_file(self, path: str, max_seconds: int = 1 << 40):
        """The program's main loop over a capture file; returns ``[second][row]`` reports."""
        nsec = min(max_seconds, os.path.getsize(path) // (8 * self.n_in)) if os.path.exists(path) else 0
        rep = (L.twx_rx_report * max(1, nsec * self.n_rows))()
        done = C.c_int64()
        self._check(self._lib.twx_rx_file(self._h, os.fsencode(path), max_seconds, rep, nsec, C.byref(done)))
        return [[rep[s * self.n_rows + i] for i in range(self.n_rows)] for s in range(done.value)]

    def powers(self):
        """Received power of physical channels A and B in the last second (V^2; 0 for a channel no row listens to)."""
        p = (C.c_double * 2)()
        self._check(self._lib.twx_rx_powers(self._h, p))
        return p[0], p[1]

    def console_line(self, i: int, report) -> str:
        """The line the program prints for channel ``i`` after a second (rxcomplex.cpp:806-831)."""
        buf = C.create_string_buffer(512)
        n = self._lib.twx_rx_console_line(self._h, i, C.byref(report), buf, len(buf))
        if n < 0:
            raise L.TwxError(n, "twx_rx_console_line")
        return buf.value.decode()

    def stream_dev(self, physical_channel: int) -> int:
        return int(self._lib.twx_rx_stream_dev(self._h, physical_channel) or 0)


def main(argv=None):
    ap = argparse.ArgumentParser(description="rxcomplex drop-in: capture file + sdr.param -> .dat rows (experiments/231001_DLL_PLL)")
    ap.add_argument("data", nargs="?", default="./data.bin")
    ap.add_argument("param", nargs="?", default="sdr.param")
    ap.add_argument("--codes", default=".", help="directory of 0.bin, 1.bin (SDRcode reads <pn-100>.bin)")
    ap.add_argument("--out", default=".", help="where the .dat files and rxcomplex.log are appended")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=int, default=1 << 40)
    ap.add_argument("--real", action="store_true", help="rx.cpp instead of rxcomplex.cpp: real samples, S rows (SIC), rxreal.log")
    a = ap.parse_args(argv)
    print(a.data)
    rows = parse_param(a.param)
    with Receiver(rows, code_dir=a.codes, out_dir=a.out, seed=a.seed, real=a.real) as rx:
        for s, reps in enumerate(rx.run_file(a.data, a.seconds)):
            for i, r in enumerate(reps):
                sys.stdout.write(rx.console_line(i, r))                  # the program's own lines (:806-831)


if __name__ == "__main__":
    main()
