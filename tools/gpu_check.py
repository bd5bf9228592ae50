#!/usr/bin/env python3
"""Quick diagnostics of the HIP path against the oracle (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_numpy, lfsr_chips_device
from oracle import twstft_oracle as orc

fs = 5e6
def case(bitlen, taps, nchips, df, seed, estimate=True, nwin=2):
    chips = prn.lfsr_chips(bitlen, taps, nchips)
    n = 2 * nchips
    chans = [synth.SynthParams(delay_q8=(n // 3 + 1157) * 256, fstep=synth.fstep_for_df(df, fs), phi0=1 << 29, amp=300,
                               noise_gain=synth.noise_gain_for_sigma(500.0), seed=seed, stream=0),
             synth.SynthParams(delay_q8=(n // 5) * 256, fstep=0, phi0=0, amp=3000,
                               noise_gain=synth.noise_gain_for_sigma(100.0), seed=seed, stream=1)]
    raw = synth.synth_capture(n * nwin, chips, 2, chans)
    t0 = time.time()
    cor = Correlator(chips, fs=fs, Nint=1)
    print(f"N={n}: N1={cor.info.n1} N2={cor.info.n2} batch={cor.info.batch} create {time.time()-t0:.2f}s")
    # FFT check
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    X = cor.fft(x); Xr = np.fft.fft(x)
    print("  fft rel err", np.abs(X - Xr).max() / np.abs(Xr).max())
    cs = cor.code_spectrum(); csr = orc.make_fcode(orc.make_code(chips, 2))
    print("  code spectrum rel err", np.abs(cs - csr).max() / np.abs(csr).max())
    band = band_numpy(fs, n)
    res = cor.ranging(raw, n_channels=2, channels=(0, 1), band=band)
    ref = orc.ranging(raw, chips, fs=fs, Nint=1, n_channels=2, band="numpy")
    for c in (0, 1):
        for w in range(nwin):
            g, o = res[c][w], ref[c][w]
            print(f"  ch{c} w{w}: indice {g.indice} vs {o['indice']}  corr {g.correction:.6f} vs {o['correction']:.6f}  "
                  f"|xval| rel {abs(abs(g.xval)-abs(o['xval']))/abs(o['xval']):.2e} xval rel {abs(g.xval-o['xval'])/abs(o['xval']):.2e} "
                  f"df {g.df:.4f} vs {o['df']:.4f} SNRr {g.SNRr:.5e} vs {o['SNRr']:.5e} SNRi {g.SNRi:.4e} vs {o['SNRi']:.4e} "
                  f"P {g.puissance:.6e} vs {o['puissance']:.6e} pn {g.puissancenoise:.6e} vs {o['puissancenoise']:.6e}")
    # full map
    d = orc.deinterleave(raw[:n], 2, 0); d = d - d.mean()
    y = d * np.exp(-2j * np.pi * ref[0][0]['df'] * np.arange(n) / fs)
    zr = orc.xcorr_interp(np.fft.fft(y), csr, 1)
    z = cor.xcorr_map(raw[:n], ref[0][0]['df'], n_channels=2, channel=0)
    print("  map rel err (max/peak)", np.abs(z - zr).max() / np.abs(zr).max(), "argmax", int(np.abs(z).argmax()), int(np.abs(zr).argmax()))
    cor.close()

print("lfsr device == host:", np.array_equal(lfsr_chips_device(17, 9, 100000), prn.lfsr_chips(17, 9, 100000)))
case(14, 43, 10000, -1210.5, 12)
case(17, 9, 100000, 1780.75, 14)
case(19, 39, 500000, 1780.75, 15, nwin=1)
if len(sys.argv) > 1:
    case(22, 3, 2500000, 1780.75, 7, nwin=1)
