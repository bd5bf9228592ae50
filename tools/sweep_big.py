#!/usr/bin/env python3
"""Randomised HIP-vs-oracle sweep at the production code lengths (run on the GPU box): python tools/sweep_big.py [nwin]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_numpy
from oracle import twstft_oracle as orc
FS = 5e6
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(7)
bad = 0; tot = 0; worst = 0.0
for bitlen, taps, nchips in ((17, 9, 100000), (19, 39, 500000)):
    chips = prn.lfsr_chips(bitlen, taps, nchips); n = 2 * nchips
    code = orc.make_code(chips, 2); fcode = orc.make_fcode(code); freq = orc.freq_axis(FS, n)
    k = orc.band_numpy(freq); temps = np.arange(n) / FS; band = band_numpy(FS, n)
    nw = nwin if nchips == 100000 else max(8, nwin // 5)
    raws = []
    for w in range(nw):
        p = synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256 + int(rng.integers(0, 256)), fstep=synth.fstep_for_df(float(rng.uniform(-7000, 7000)), FS),
                              phi0=int(rng.integers(0, 2 ** 32)), amp=int(rng.choice([0, 10, 40, 200])), noise_gain=synth.noise_gain_for_sigma(float(rng.choice([100.0, 600.0]))),
                              seed=int(rng.integers(1, 10 ** 6)))
        raws.append(synth.synth_channel(n, chips, 2, p))
    with Correlator(chips, fs=FS, Nint=1) as cor:
        got = cor.process(np.concatenate(raws), 1, 0, band=band)
    t = time.time()
    for w, g in enumerate(got):
        d = orc.deinterleave(raws[w], 1, 0); d = d - d.mean()
        o = orc.processing(d, k, freq, temps, fcode, code, Nint=1, fs=FS)
        tot += 1
        if g.indice != o["indice"] or abs(g.df - o["df"]) > 1e-9:
            bad += 1; print("MISMATCH", nchips, w, g.indice, o["indice"], g.df, o["df"])
        else:
            worst = max(worst, abs(abs(g.xval) - abs(o["xval"])) / abs(o["xval"]))
    print(f"N={n}: {nw} windows, oracle {time.time()-t:.1f} s")
print(f"{tot} cases, {bad} mismatches, worst |xval| relative error {worst:.2e}")
