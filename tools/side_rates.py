#!/usr/bin/env python3
"""Throughput of the secondary entry points at BASELINE.json sizes (run on the GPU box).

  configs[2]: full delay x Doppler CAF, +-5 kHz at 1 Hz over one 1-s window (10 001 bins, N = 5e6)
  tracked 40-ms loop (N = 200 000, 2-s chunks) and search_df's 1e7-point squared spectrum
  fp64 chain (configs[4]'s tolerance leg)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual
from amaranth_twstft_amd.tracked import TrackedRanging

FS = 5e6


def caf():
    chips = prn.lfsr_chips(22, 3, 2_500_000)
    n = 5_000_000
    p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1781.0, FS), phi0=1, amp=200,
                          noise_gain=synth.noise_gain_for_sigma(400.0), seed=7)
    raw = synth.synth_channel(n, chips, 2, p)
    with Correlator(chips, fs=FS, Nint=0) as cor:
        cor.caf_bins(raw, -8, 8)
        t = time.time()
        pk, lag = cor.caf_bins(raw, -5000, 5000)
        dt = time.time() - t
    b = int(np.argmax(pk))
    print(f"CAF 10001 bins x 5e6 samples: {dt:.3f} s  ({10001 * n / dt / 1e9:.1f} Gsample-bins/s, {n / dt / 1e6:.1f} Msample/s of input) "
          f"peak bin {b - 5000} lag {lag[b]}")


def tracked():
    nchips, n = 100000, 200000
    chips = prn.lfsr_chips(17, 9, nchips)
    p = synth.SynthParams(delay_q8=123456 * 256, fstep=synth.fstep_for_df(12.0, FS), phi0=9, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=21)
    raw = synth.synth_channel(n * 50 * 6 + n, chips, 2, p)
    with TrackedRanging(chips, fs=FS, Nint=1) as tr:
        t = time.time(); kb = tr.search_df(raw.reshape(-1)[:2 * tr.L]); t_search = time.time() - t
        t = time.time(); out = tr.run(raw, kbon=kb); dt = time.time() - t
    ns = len(out["indice1"]) * n
    print(f"tracked loop: search_df {t_search:.3f} s (kbon {kb}); {len(out['indice1'])} codes in {dt:.3f} s = {ns / dt / 1e6:.0f} Msample/s "
          f"(host-fed, {out['batches']} batches, moved {out['moved']})")


def f64():
    chips = prn.lfsr_chips(22, 3, 2_500_000)
    n = 5_000_000
    p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=1, amp=200,
                          noise_gain=synth.noise_gain_for_sigma(400.0), seed=7)
    raw = np.tile(synth.synth_channel(n, chips, 2, p).reshape(-1), 8)
    for prec in ("f32", "f64"):
        with Correlator(chips, fs=FS, Nint=1, precision=prec) as cor:
            cor.process(raw, 1, 0, band=band_godual(FS, n))
            t = time.time(); r = cor.process(raw, 1, 0, band=band_godual(FS, n)); dt = time.time() - t
        print(f"{prec}: 8 windows host-fed {dt:.3f} s = {8 * n / dt / 1e6:.0f} Msample/s indice {r[0].indice} |xval| {abs(r[0].xval):.9e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["caf", "tracked", "f64"]
    for w in which:
        globals()[w]()
