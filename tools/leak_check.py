import sys; sys.path.insert(0,'.')
import torch, numpy as np
from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_numpy
chips=prn.lfsr_chips(17,9,100000); n=200000
p=synth.SynthParams(delay_q8=777*256, fstep=0, phi0=1, amp=300, noise_gain=synth.noise_gain_for_sigma(300.0), seed=1)
raw=synth.synth_channel(n*4, chips, 2, p)
free0=None
for it in range(40):
    with Correlator(chips, fs=5e6, Nint=1) as cor:
        r=cor.process(raw,1,0,band=band_numpy(5e6,n))
        cor.xcorr_map(raw[:n],0.0); cor.caf_bins(raw[:n],-3,3)
    f,t=torch.cuda.mem_get_info()
    if it==2: free0=f
    if it%13==0: print(it, f>>20)
print('leak MB:', (free0-f)>>20)
