#!/usr/bin/env python3
"""Rate of the direct sliding dot-product correlator at the tracking stage's sizes (rxcomplex.cpp:593-605:
24 code periods x 57 lags per second of signal) — run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn, synth, tracking
FS = 5e6
nchips, nobs, ncodes, nlag = 100000, 200000, 24, 28
chips = prn.lfsr_chips(17, 9, nchips)
p = synth.SynthParams(delay_q8=5 * 256, fstep=synth.fstep_for_df(3.0, FS), phi0=9, amp=300, noise_gain=synth.noise_gain_for_sigma(500.0), seed=2)
raw = synth.synth_channel(nobs * (ncodes + 1), chips, 2, p)
rep = tracking.prn_sampling(nobs, 2.0 * chips.astype(np.float64) - 1.0, 2.5e6, FS, 0.0)
res = tracking.sliding_dot(raw, rep, nobs, ncodes, nlag, pt=0, ff=3.0 / FS)
t = time.time()
for _ in range(5):
    res = tracking.sliding_dot(raw, rep, nobs, ncodes, nlag, pt=0, ff=3.0 / FS)
dt = (time.time() - t) / 5
cor, phi = tracking.get_cor_and_phi(res)
pk, hrc = tracking.hrc_delay(cor, nlag)
macs = ncodes * (2 * nlag + 1) * nobs
print(f"sliding_dot {ncodes}x{2*nlag+1} lags x {nobs} samples: {dt*1e3:.2f} ms host-fed = {ncodes*nobs/dt/1e6:.0f} Msample/s, "
      f"{macs/dt/1e9:.0f} G complex-real MAC/s; peak lags {sorted(set(pk.tolist()))}")
