import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from amaranth_twstft_amd.correlator import Correlator, band_godual
from amaranth_twstft_amd import synth
from tests.helpers import chips_for
from tests.test_gpu_configs import _synth_dev
dev = torch.device("cuda", 0)
fs, sps, n = 70e6, 28, 70_000_000
chips = chips_for(22, 57, 2_500_000)
p = synth.SynthParams(delay_q8=18_364_717 * 256, fstep=synth.fstep_for_df(3.25, fs), phi0=99, amp=2500, noise_gain=synth.noise_gain_for_sigma(2500.0), seed=401)
nw = 4
wide = torch.empty((nw * n, 2), dtype=torch.int16, device=dev)
_synth_dev(wide, nw * n, torch.from_numpy(chips).to(dev), 2_500_000, sps, [p])
torch.cuda.synchronize()
band = band_godual(fs, n)
with Correlator(chips, fs=fs, sps=sps, Nint=1, profile=True) as cor:
    print(cor.info.n1, cor.info.n2, cor.info.batch)
    cor.process_dev(wide.data_ptr(), nw, band=band)
    t = time.time(); g = cor.process_dev(wide.data_ptr(), nw, band=band); dt = time.time() - t
    print("windows", nw, "s/window", dt / nw, "Gsample/s", nw * n / dt / 1e9, [x.indice for x in g])
    try:
        print(cor.profile())
    except Exception as e:
        print("no profile", e)
