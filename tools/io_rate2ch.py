#!/usr/bin/env python3
"""2-channel capture file, both channels per window: two single-channel passes vs the all-channel mode (GPU box)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn
from amaranth_twstft_amd.correlator import Correlator, band_godual
NCH = 2500000; N = 2 * NCH; NW = int(sys.argv[1]) if len(sys.argv) > 1 else 96
chips = prn.lfsr_chips(22, 3, NCH)
rng = np.random.default_rng(1)
code = np.repeat(chips.astype(np.int16), 2) * 2 - 1
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    path = os.path.join(td, "1670000000.bin")
    block = []
    for w in range(4):
        x1 = (200 * np.roll(code, 1311765 - w)).astype(np.int16); x2 = (3000 * np.roll(code, 3626553)).astype(np.int16)
        iq = np.stack([x1, np.zeros_like(x1), x2, np.zeros_like(x2)], axis=1) + rng.integers(-400, 400, (N, 4), dtype=np.int16)
        block.append(iq.astype(np.int16))
    block = np.concatenate(block)
    with open(path, "wb") as f:
        for _ in range(NW // 4):
            block.tofile(f)
    band = band_godual(5e6, N)
    with Correlator(chips, fs=5e6, Nint=1) as cor:
        cor.process_file(path, 2, -1, band=band, max_windows=8)
        t = time.perf_counter(); a = [cor.process_file(path, 2, c, band=band) for c in (0, 1)]; dt_sep = time.perf_counter() - t
        t = time.perf_counter(); b = cor.process_file(path, 2, -1, band=band); dt_all = time.perf_counter() - t
    ok = all(a[c][w].indice == b[c][w].indice for c in (0, 1) for w in range(NW)) and b[0][1].indice == 3 * 1311764 and b[1][0].indice == 3 * 3626553
    cs = 2 * NW * N
    print(f"{NW} windows x 2 channels: two passes {dt_sep*1e3:.0f} ms = {cs/dt_sep/1e6:.0f} M channel-samples/s; all-channel mode {dt_all*1e3:.0f} ms = {cs/dt_all/1e6:.0f} M channel-samples/s; equal={ok}")
