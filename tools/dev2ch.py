#!/usr/bin/env python3
"""Device-resident 2-channel capture: one call per channel vs all-channel mode (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from amaranth_twstft_amd import _lib as L, prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual
FS = 5e6; NCH = 2500000; N = 2 * NCH; NW = 32
lib = L.load()
chips = prn.lfsr_chips(22, 3, NCH)
chips_dev = torch.from_numpy(chips).cuda()
iq = torch.empty((NW, N, 4), dtype=torch.int16, device="cuda")
for p in range(NW):
    par = np.array([(1311765 - p) * 256, synth.fstep_for_df(1780.75, FS), 7 * p, 200, synth.noise_gain_for_sigma(400.0), 1000 + p, 0, 0,
                    3626553 * 256, 0, 0, 3000, synth.noise_gain_for_sigma(100.0), 1000 + p, 1, 0], dtype=np.int64)
    L.check(lib.twx_synchronize) if False else None
    L.check(lib.twx_synth_capture_dev(iq[p].data_ptr(), N, 0, chips_dev.data_ptr(), NCH, 2, 2, par.ctypes.data_as(C.c_void_p), None))
torch.cuda.synchronize()
band = band_godual(FS, N)
with Correlator(chips, fs=FS, Nint=1) as cor:
    cor.process_dev(iq.data_ptr(), NW, 2, -1, band=band)
    t = time.perf_counter(); a = [cor.process_dev(iq.data_ptr(), NW, 2, c, band=band) for c in (0, 1)]; d1 = time.perf_counter() - t
    t = time.perf_counter(); b = cor.process_dev(iq.data_ptr(), NW, 2, -1, band=band); d2 = time.perf_counter() - t
ok = all(a[c][w].indice == b[c][w].indice for c in (0, 1) for w in range(NW)) and b[0][3].indice == 3 * (1311765 - 3) and b[1][0].indice == 3 * 3626553
print(f"device-resident {NW} windows x 2 ch: per-channel calls {2*NW*N/d1/1e9:.1f} G ch-samples/s, all-channel {2*NW*N/d2/1e9:.1f} G ch-samples/s, equal={ok}")
