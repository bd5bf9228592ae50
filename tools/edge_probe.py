#!/usr/bin/env python3
"""Peak neighbours across the row boundary of the two-pass layout: windows whose correlation peak sits at q2 = 0, 1, n2-1 (and the
same for every phase rho) in fp32 and fp64, xvalm1 / xval / xvalp1 and the correction against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import synth
from amaranth_twstft_amd.correlator import Correlator
from oracle import twstft_oracle as orc
from tests.helpers import chips_for

FS = 5e6
bad = 0
for (bitlen, taps, nchips) in ((14, 57, 10000), (13, 27, 5000), (15, 17, 25000)):
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    code = orc.make_code(chips, 2); fcode = orc.make_fcode(code); temps = np.arange(n) / FS
    for precision in ("f32", "f64"):
        for Nint in (0, 1, 2):
            with Correlator(chips, fs=FS, Nint=Nint, precision=precision) as cor:
                n2 = int(cor.info.n2)
                delays = sorted(set([0, 1, n2 - 1, n2, n2 + 1, 17 * n2 - 1, 17 * n2, 17 * n2 + 1, n - 1, n - n2, 5 * n2 + 7]))
                for d in delays:
                    for frac in (0, 90, 170):                     # sub-sample delay in 1/256: moves the peak between the phases
                        p = synth.SynthParams(delay_q8=d * 256 + frac, fstep=synth.fstep_for_df(1234.5, FS), phi0=77, amp=1500,
                                              noise_gain=synth.noise_gain_for_sigma(200.0), seed=d + 1)
                        raw = synth.synth_channel(n, chips, 2, p)
                        g = cor.process(raw, 1, 0, df=1234.5)[0]
                        x = orc.deinterleave(raw, 1, 0); x = x - x.mean()
                        o = orc.processing(x, None, None, temps, fcode, code, Nint=Nint, fs=FS, df=1234.5)
                        tol = 2e-6 * abs(o["xval"])
                        okk = g.indice == o["indice"] and abs(g.xvalm1 - o["xvalm1"]) <= tol and abs(g.xvalp1 - o["xvalp1"]) <= tol and abs(g.correction - o["correction"]) <= 2e-4 \
                            and abs(g.SNRr - o["SNRr"]) <= 1e-3 * max(o["SNRr"], o["SNRi"]) + 1e-30
                        if not okk:
                            bad += 1
                            print("MISMATCH n", n, "n2", n2, precision, "Nint", Nint, "delay", d, frac, "indice", g.indice, o["indice"], "m1", abs(g.xvalm1), abs(o["xvalm1"]),
                                  "p1", abs(g.xvalp1), abs(o["xvalp1"]), "corr", g.correction, o["correction"], flush=True)
print("mismatches:", bad)
