"""Create / use / destroy cycles of every context-owning object against the free device memory (run on the GPU box):
Correlator (chain, full map, CAF with its persistent bin buffer), TrackedRanging (twx_tracked_*: sample buffer, pinned
staging, inner context), Acquisition (one-call sweep, decimated form), the tracking epoch's scratch, twx_ctx_alloc without a
matching free, the multi-GPU driver (twx_multi_*: contexts, worker threads, gather buffers, the RCCL world of one) and the receiver
program (twx_rx_*: fp64 work contexts of the set-up, acquisition contexts, replica, streams).  Prints the drift in MB after the
warm-up cycles: it must be 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from amaranth_twstft_amd import acquisition, prn, synth, tracking
from amaranth_twstft_amd.correlator import Correlator, band_numpy
from amaranth_twstft_amd.tracked import TrackedRanging
from amaranth_twstft_amd.multi import MultiCorrelator
from amaranth_twstft_amd import receiver

chips = prn.lfsr_chips(17, 9, 100000); n = 200000
p = synth.SynthParams(delay_q8=777 * 256, fstep=synth.fstep_for_df(40.0, 5e6), phi0=1, amp=300, noise_gain=synth.noise_gain_for_sigma(300.0), seed=1)
raw = synth.synth_channel(n * 12, chips, 2, p)
smp = torch.randn((1 << 21, 2), dtype=torch.float32, device="cuda")
iq = torch.from_numpy(raw).cuda()
rep = torch.from_numpy(tracking.prn_sampling(n, 2.0 * chips - 1.0, 2.5e6, 5e6)).cuda()
free0 = None
for it in range(24):
    with Correlator(chips, fs=5e6, Nint=1) as cor:
        cor.process(raw, 1, 0, band=band_numpy(5e6, n))
        cor.xcorr_map(raw[:n], 0.0); cor.caf_bins(raw[:n], -70, 70)
        cor._lib.twx_ctx_alloc(cor._h, 64 << 20)                   # never freed by the caller: twx_destroy must
        st = dict(fc=40.0, pt=0, last_phi=0.0, psbb=1.0, duration=n / 5e6, fs=5e6)
        tracking.track_epoch_dev(cor, iq.data_ptr(), n * 12, rep.data_ptr(), n, 11, 28, st, scale=1.0)
    for mode in ("ranging", "lo"):
        with TrackedRanging(chips, fs=5e6, Nint=1, ls_samples=4 * n, mode=mode) as tr:
            tr.run(raw)
    with MultiCorrelator(chips, [0, 0, 0], fs=5e6, Nint=1) as m:
        m.process(raw, 1, 0, band=band_numpy(5e6, n))
        m.process_dev([iq.data_ptr()] * 3, 4, band=band_numpy(5e6, n))
    if it % 4 == 0:
        with MultiCorrelator(chips, [0], fs=5e6, Nint=1, rccl=True) as m:
            m.process(raw, 1, 0, band=band_numpy(5e6, n))
        with receiver.Receiver([receiver.make_row("A", 100, 186.0, 1000.0, 256.0, -18.0, code=chips)], fs_in=2.5e6) as rx:   # 2.5 Msps: nobs = 200 000, nfft = 2^19 (plans that exist), half the work of a 5-Msps second
            rx.second(np.zeros((2_500_000, 4), dtype=np.int16))
        # the real-sample program with an 'S' row (rx.cpp): own stream, interference records, the cleaned stream
        with receiver.Receiver([receiver.make_row("A", 100, 186.0, 1000.0, 256.0, -18.0, code=chips),
                                receiver.make_row("A", 101, 186.0, 1000.0, 256.0, -18.0, code=chips[::-1].copy(), mode="S")], fs_in=5e6, real=True) as rx:
            rx.second(np.zeros((5_000_000, 4), dtype=np.int16))
    a = acquisition.Acquisition(1 - 2 * chips.astype(np.int64), 2.5e6, 10e6, 400000, dec_a=1 + it % 2, max_batch=16)
    a.acquire(smp.data_ptr(), 0, 100.0, 1024.0, 256.0)
    a.close()
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    if it == 3:
        free0 = f
    if it % 6 == 0:
        print(it, f >> 20)
print('leak MB:', (free0 - f) >> 20)
