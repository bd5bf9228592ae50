#!/usr/bin/env python3
"""One kernel class of the chain looped ALONE for a few seconds (python tools/kernel_alone.py [class] [seconds] [slots]):
what does k_row_mid do when nothing else shares the GPU — time per launch, and (sampled by the calling script with
rocm-smi) board power and shader clock.  Uses TWX_OPT_DEBUG_ONLY / _REPEAT on real batch data left by a complete pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from amaranth_twstft_amd import _lib as L, prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual

NAMES = ["k_sums", "k_col_fwd_square", "k_row_band", "k_df_tables", "k_col_fwd_mix", "k_row_mid", "k_col_inv", "k_peak"]
cls = NAMES.index(sys.argv[1]) if len(sys.argv) > 1 else 5
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
zero = len(sys.argv) > 3 and sys.argv[3] == "zero"      # all-zero capture: same instruction stream, (almost) no data toggling
FS, NCH = 5e6, 2_500_000
n = 2 * NCH
lib = L.load()
chips = prn.lfsr_chips(22, 3, NCH)
nwin = 24
p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=1, amp=200, noise_gain=synth.noise_gain_for_sigma(400.0), seed=5)
chips_dev = torch.from_numpy(chips).cuda()
iq = torch.empty((nwin, n, 2), dtype=torch.int16, device="cuda")
params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
for w in range(nwin):
    params[5] = 1000 + w
    L.check(lib.twx_synth_capture_dev(iq[w].data_ptr(), n, 0, chips_dev.data_ptr(), NCH, 2, 1, params.ctypes.data_as(C.c_void_p), None))
if zero:
    iq.zero_()
torch.cuda.synchronize()
res = torch.zeros((nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device="cuda")
band = L.twx_band(*band_godual(FS, n))
with Correlator(chips, fs=FS, Nint=1) as cor:
    B = int(cor.info.batch)
    def run(nw):
        L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nw, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
        L.check(lib.twx_synchronize(cor._h), cor._h)
    run(nwin)                                           # a complete pass: every slot's batch buffers hold real data
    L.check(lib.twx_set_option(cor._h, 100, cls), cor._h)
    L.check(lib.twx_set_option(cor._h, 101, 20), cor._h)
    t = time.perf_counter(); run(B); dt = (time.perf_counter() - t) / 20      # one batch on slot 0 only
    reps = max(20, int(seconds / dt))
    L.check(lib.twx_set_option(cor._h, 101, reps), cor._h)
    t = time.perf_counter(); run(B); el = time.perf_counter() - t
    print(f"{NAMES[cls]} alone{' (all-zero capture)' if zero else ''}, 1 slot, batch {B}: {reps} launches in {el:.2f} s = {el / reps * 1e3:.4f} ms per launch")
    L.check(lib.twx_set_option(cor._h, 100, -1), cor._h)
