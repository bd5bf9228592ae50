#!/usr/bin/env python3
"""Where does the fp64 correlation map differ from the oracle's?  Full maps (twx_xcorr_map) at N = 10000, 20000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import synth
from amaranth_twstft_amd.correlator import Correlator
from oracle import twstft_oracle as orc
from tests.helpers import chips_for
FS = 5e6
for (bitlen, taps, nchips) in ((13, 27, 5000), (14, 57, 10000)):
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    code = orc.make_code(chips, 2); fcode = orc.make_fcode(code); temps = np.arange(n) / FS
    p = synth.SynthParams(delay_q8=1234 * 256, fstep=synth.fstep_for_df(0.0, FS), phi0=77, amp=1500, noise_gain=synth.noise_gain_for_sigma(200.0), seed=3)
    raw = synth.synth_channel(n, chips, 2, p)
    x = orc.deinterleave(raw, 1, 0); x = x - x.mean()
    for precision in ("f32", "f64"):
        for Nint in (0, 1):
            with Correlator(chips, fs=FS, Nint=Nint, precision=precision) as cor:
                z = cor.xcorr_map(raw, 0.0, n_channels=1, channel=0)
                n1, n2 = int(cor.info.n1), int(cor.info.n2)
            zr = orc.xcorr_interp(np.fft.fft(x), fcode, Nint)
            R = 2 * Nint + 1
            err = np.abs(z - zr)
            badm = np.nonzero(err > 1e-5 * np.abs(zr).max())[0]
            q = badm // R
            print(n, n1, n2, precision, "Nint", Nint, "bad", badm.size, "of", z.size, "q2 set", sorted(set((q % n2).tolist()))[:12], "q1 count", len(set((q // n2).tolist())),
                  "rho", sorted(set((badm % R).tolist())), "max err rel", float(err.max() / np.abs(zr).max()))
