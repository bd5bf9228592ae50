#!/usr/bin/env python3
"""PCIe- and file-inclusive rate of the file-in/results-out path; never bench.py's `value`."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn
from amaranth_twstft_amd.correlator import Correlator, band_godual
NCH = 2500000; N = 2 * NCH; NW = int(sys.argv[1]) if len(sys.argv) > 1 else 48
chips = prn.lfsr_chips(22, 3, NCH)
rng = np.random.default_rng(1)
code = np.repeat(chips.astype(np.int16), 2) * 2 - 1
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    path = os.path.join(td, "1670000000.bin")
    with open(path, "wb") as f:
        for w in range(NW):
            x = (200 * np.roll(code, 1311765 - w)).astype(np.int16)
            iq = np.stack([x, np.zeros_like(x)], axis=1) + rng.integers(-400, 400, (N, 2), dtype=np.int16)
            iq.astype(np.int16).tofile(f)
    with Correlator(chips, fs=5e6, Nint=1) as cor:
        band = band_godual(5e6, N)
        cor.process_file(path, n_channels=1, channel=0, band=band, max_windows=8)      # warm
        t0 = time.perf_counter()
        res = cor.process_file(path, n_channels=1, channel=0, band=band)
        dt = time.perf_counter() - t0
        raw = np.fromfile(path, dtype=np.int16)
        t1 = time.perf_counter()
        res2 = cor.process(raw, n_channels=1, channel=0, band=band)
        dt2 = time.perf_counter() - t1
    ok = all(r.indice == 3 * (1311765 - w) for w, r in enumerate(res))
    print(f"process_file: {NW} windows in {dt*1e3:.1f} ms = {NW*N/dt/1e6:.0f} Msample/s ({NW*N*4/dt/1e9:.1f} GB/s of int16), lags ok={ok}")
    print(f"process (host buffer, pageable): {dt2*1e3:.1f} ms = {NW*N/dt2/1e6:.0f} Msample/s")
