#!/usr/bin/env python3
"""FIR front end of configs[4] on one channel of a TWO-channel 70 Msps capture against a one-channel capture (staging path)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amaranth_twstft_amd import frontend
from amaranth_twstft_amd.correlator import Correlator
dev = torch.device("cuda", 0)
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
dec, nout = 14, 5_000_000
n_in = (nout - 1) * dec + taps.size
with Correlator(lfsr=(14, 43, 10000), fs=5e6) as cor:
    for nch in (1, 2):
        x = torch.randint(-3000, 3000, (n_in, 2 * nch), dtype=torch.int16, device=dev)
        y16 = torch.empty((nout, 2), dtype=torch.int16, device=dev)
        for ch in range(nch):
            f = lambda: cor.fir_decimate_dev(x.data_ptr(), n_in, taps, dec, y16.data_ptr(), None, n_channels=nch, channel=ch)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.6:
                for _ in range(10): f()
                cor.synchronize()
            t = time.perf_counter()
            for _ in range(20): f()
            cor.synchronize()
            dt = (time.perf_counter() - t) / 20
            print(json.dumps({"nch": nch, "channel": ch, "ms": round(dt * 1e3, 4), "TFLOP_s": round(nout * taps.size * 4 / dt / 1e12, 1)}))
