#!/usr/bin/env python3
"""Device-resident throughput of processing(d,k) for the short reference codes (run on the GPU box):
how the windows-per-launch batch size amortises launch overhead when a window is only 10^4..10^6 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from amaranth_twstft_amd import _lib as L, prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual

FS = 5e6
lib = L.load()
for bitlen, taps, nchips in ((14, 43, 10000), (17, 9, 100000), (19, 39, 500000)):
    n = 2 * nchips
    chips = prn.lfsr_chips(bitlen, taps, nchips)
    nwin = max(64, min(4096, int(160e6 // n)))
    p = synth.SynthParams(delay_q8=(n // 3) * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=1, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=5)
    raw = torch.from_numpy(synth.synth_channel(n * nwin, chips, 2, p).reshape(-1)).cuda()
    res = torch.zeros((nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device="cuda")
    band = L.twx_band(*band_godual(FS, n))
    for mb in (0, 16, 64, 256, 1024):
        if mb > nwin:
            continue
        with Correlator(chips, fs=FS, Nint=1, max_batch=mb) as cor:
            def run():
                L.check(lib.twx_process_windows_dev(cor._h, raw.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
                L.check(lib.twx_synchronize(cor._h), cor._h)
            run()
            t = time.time()
            for _ in range(3):
                run()
            dt = (time.time() - t) / 3
            print(f"N={n:8d} nwin={nwin:5d} max_batch={mb:5d} (B={cor.info.batch:4d}): {nwin * n / dt / 1e9:6.2f} Gsample/s")
