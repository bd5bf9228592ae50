"""GPU box: throughput of the B210-era two-channel interleaved capture format ([I1 Q1 I2 Q2] per sample, BASELINE.json configs[3]'s
literal layout: 600 windows x 2 channels) in all-channel mode against the one-channel-per-file layout, per CHANNEL-sample."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import _lib as L, prn, synth  # noqa: E402
from amaranth_twstft_amd.correlator import ALL_CHANNELS, Correlator, band_godual  # noqa: E402

N, NCHIPS, FS = 5_000_000, 2_500_000, 5e6
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda", 0)
lib = L.load()
chips = prn.lfsr_chips(22, 3, NCHIPS)
cdev = torch.from_numpy(chips).to(dev)
chans = [synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=0, amp=200, noise_gain=synth.noise_gain_for_sigma(400.0), seed=7, stream=0),
         synth.SynthParams(delay_q8=3626553 * 256, fstep=0, phi0=0, amp=3000, noise_gain=synth.noise_gain_for_sigma(100.0), seed=7, stream=1)]
params = np.concatenate([np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64) for p in chans])
two = torch.empty((nwin * N, 4), dtype=torch.int16, device=dev)
L.check(lib.twx_synth_capture_dev(two.data_ptr(), nwin * N, 0, cdev.data_ptr(), NCHIPS, 2, 2, params.ctypes.data_as(C.c_void_p), None))
one = two[:, 0:2].contiguous()
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
res = torch.zeros((2 * nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
out = {}
with Correlator(chips, fs=FS, Nint=1) as cor:
    def run(ptr, nch, ch, reps=3):
        L.check(lib.twx_process_windows_dev(cor._h, ptr, nwin, nch, ch, C.byref(band), None, res.data_ptr()), cor._h)
        L.check(lib.twx_synchronize(cor._h), cor._h)
        t = time.perf_counter()
        for _ in range(reps):
            L.check(lib.twx_process_windows_dev(cor._h, ptr, nwin, nch, ch, C.byref(band), None, res.data_ptr()), cor._h)
        L.check(lib.twx_synchronize(cor._h), cor._h)
        return (time.perf_counter() - t) / reps
    t1 = run(one.data_ptr(), 1, 0)
    out["one_channel_file"] = nwin * N / t1 / 1e9
    t2 = run(two.data_ptr(), 2, 0)
    out["two_channel_file_one_channel"] = nwin * N / t2 / 1e9
    t3 = run(two.data_ptr(), 2, ALL_CHANNELS)
    out["two_channel_file_all_channels"] = 2 * nwin * N / t3 / 1e9
    arr = (L.twx_result * (2 * nwin)).from_buffer_copy(res.cpu().numpy().tobytes())
    out["lags"] = [int(arr[0].indice0), int(arr[1].indice0)]
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
