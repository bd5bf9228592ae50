"""GPU box: randomised parity sweep at the FULL window length (N = 5e6, the 625 x 8000 plan) — seeded windows from a strong
signal down to pure noise (where the arg-max over the 1.5e7-point map is decided among noise peaks): integer lag and carrier
bin of the device against the fp64 oracle on the same samples.  One JSON line.    python tools/sweep_5m.py [windows] [seed]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import prn, synth  # noqa: E402
from amaranth_twstft_amd.correlator import Correlator, band_godual  # noqa: E402
from oracle import twstft_oracle as orc  # noqa: E402
from tests.test_gpu_configs import _synth_dev  # noqa: E402

N, NCHIPS, FS = 5_000_000, 2_500_000, 5e6
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261002)
dev = torch.device("cuda", 0)
chips = prn.lfsr_chips(22, 3, NCHIPS)
cdev = torch.from_numpy(chips).to(dev)
iq = torch.empty((nwin, N, 2), dtype=torch.int16, device=dev)
params = []
for w in range(nwin):
    amp = int(rng.choice([0, 0, 10, 40, 200]))
    sigma = float(rng.choice([100.0, 400.0, 3000.0]))
    df = float(rng.uniform(-9000, 9000))
    p = synth.SynthParams(delay_q8=int(rng.integers(0, N)) * 256 + int(rng.integers(0, 256)), fstep=synth.fstep_for_df(df, FS),
                          phi0=int(rng.integers(0, 2 ** 32)), amp=amp, noise_gain=synth.noise_gain_for_sigma(sigma), seed=int(rng.integers(1, 10 ** 6)))
    _synth_dev(iq[w], N, cdev, NCHIPS, 2, [p])
    params.append((amp, sigma, round(df, 2)))
torch.cuda.synchronize()
band = band_godual(FS, N)
with Correlator(chips, fs=FS, Nint=1) as cor:
    got = cor.process_dev(iq.data_ptr(), nwin, band=band)
with Correlator(chips, fs=FS, Nint=1, precision="f64") as cor64:      # round 5: the complex-double chain on the same windows
    got64 = cor64.process_dev(iq.data_ptr(), nwin, band=band)
code = orc.make_code(chips, 2)
fcode = orc.make_fcode(code)
freq = orc.freq_axis(FS, N)
k = np.arange(band[0], band[1] + 1)
temps = np.arange(N) / FS
mism, relmax, mism64, relmax64 = [], 0.0, [], 0.0
t0 = time.time()
for w in range(nwin):
    d = orc.deinterleave(iq[w].cpu().numpy(), 1, 0)
    d = d - d.mean()
    o = orc.processing(d, k, freq, temps, fcode, code, Nint=1, fs=FS)
    g = got[w]
    if g.indice != o["indice"] or abs(g.df - o["df"]) > 1e-9:
        mism.append((w, params[w], g.indice, o["indice"], g.df, o["df"]))
    else:
        relmax = max(relmax, abs(abs(g.xval) - abs(o["xval"])) / abs(o["xval"]))
    g = got64[w]
    if g.indice != o["indice"] or abs(g.df - o["df"]) > 1e-9:
        mism64.append((w, params[w], g.indice, o["indice"], g.df, o["df"]))
    else:
        relmax64 = max(relmax64, abs(abs(g.xval) - abs(o["xval"])) / abs(o["xval"]))
print(json.dumps({"windows": nwin, "n": N, "mismatches": mism, "max_rel_peak_error": relmax, "f64_mismatches": mism64, "f64_max_rel_peak_error": relmax64, "noise_only_windows": sum(1 for p in params if p[0] == 0),
                  "oracle_seconds": round(time.time() - t0, 1)}))
