#!/usr/bin/env python3
"""Rates of the kernels beside the FFT chain at BASELINE.json sizes, device-resident (run on the GPU box):

  fir      70 Msps -> 5 Msps front end of configs[4]: 577 taps, decimate by 14, 1 s of samples per call (twx_fir_decimate_dev)
  sliding  direct sliding dot product of the tracking stage (a12): 24 code periods of 400 000 samples, +-28 lags
           (experiments/231001_DLL_PLL/rxcomplex.cpp:593-605) (twx_sliding_dot_dev)
  acq      acquisition sweep of rxcomplex.cpp:534-567 at sdr.param sizes: 513 + 24 trial carriers, two 2^20-point transforms each
  interp   short2double x2 interpolation, 5e6 -> 1e7 samples

Prints one JSON line per kernel with the algorithmic bytes / flops, the time per call (wall clock around N calls and one
twx_synchronize) and both rooflines: HBM (8 TB/s) and fp32 vector (157.3 TFLOP/s; SURVEY.md §8d puts a12 and the FIR above
the ridge).  Under rocprofv3 --kernel-trace --stats the same run gives the per-kernel durations in profiles/.
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import frontend, prn, synth
from amaranth_twstft_amd.correlator import Correlator

HBM, VEC = 8000.0, 157.3
dev = torch.device("cuda", 0)


def timed(fn, sync, reps):
    t0 = time.perf_counter()                    # warm-up: the clocks need some tenths of a second of load to leave the idle state
    while time.perf_counter() - t0 < 0.6:
        for _ in range(10):
            fn()
        sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t) / reps


def fir():
    taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
    dec, nout = 14, 5_000_000
    n_in = (nout - 1) * dec + taps.size
    x = torch.randint(-3000, 3000, (n_in, 2), dtype=torch.int16, device=dev)
    y16 = torch.empty((nout, 2), dtype=torch.int16, device=dev)
    yf = torch.empty((nout, 2), dtype=torch.float32, device=dev)
    with Correlator(lfsr=(14, 43, 10000), fs=5e6) as cor:
        for name, o16, of in (("int16 out", y16.data_ptr(), None), ("int16 + float out", y16.data_ptr(), yf.data_ptr())):
            dt = timed(lambda: cor.fir_decimate_dev(x.data_ptr(), n_in, taps, dec, o16, of), cor.synchronize, 20)
            byts = n_in * 4 + nout * (4 + (8 if of else 0))
            flops = nout * taps.size * 2 * 2
            print(json.dumps({"kernel": "k_fir_poly", "case": f"70 Msps x 1 s -> 5 Msps, {taps.size} taps, dec {dec}, {name}", "ms": round(dt * 1e3, 4),
                              "input_Gsample_s": round(n_in / dt / 1e9, 1), "algorithmic_bytes": byts, "GB_s": round(byts / dt / 1e9, 1),
                              "frac_hbm": round(byts / dt / 1e9 / HBM, 4), "flops": flops, "TFLOP_s": round(flops / dt / 1e12, 1),
                              "frac_fp32_vector": round(flops / dt / 1e12 / VEC, 4)}))


def sliding():
    nobs, ncodes, nlag = 400_000, 24, 28
    n = nobs * ncodes + 64
    x = torch.randint(-3000, 3000, (n, 2), dtype=torch.int16, device=dev)
    rep = (torch.randint(0, 2, (nobs,), device=dev).float() * 2 - 1).contiguous()
    out = torch.empty((ncodes, 2 * nlag + 1, 2), dtype=torch.float64, device=dev)
    with Correlator(lfsr=(14, 43, 10000), fs=5e6) as cor:
        dt = timed(lambda: cor.sliding_dot_dev(x.data_ptr(), n, rep.data_ptr(), nobs, ncodes, nlag, out.data_ptr(), ff=1.234e-5, scale=1.0 / 32768),
                   cor.synchronize, 50)
    byts = nobs * ncodes * 4 + nobs * 4
    flops = nobs * ncodes * (2 * nlag + 1) * 4
    print(json.dumps({"kernel": "k_sliding_dot<28>", "case": f"{ncodes} codes x {nobs} samples, +-{nlag} lags", "ms": round(dt * 1e3, 4),
                      "Gsample_s": round(nobs * ncodes / dt / 1e9, 1), "algorithmic_bytes": byts, "GB_s": round(byts / dt / 1e9, 1),
                      "frac_hbm": round(byts / dt / 1e9 / HBM, 4), "flops": flops, "TFLOP_s": round(flops / dt / 1e12, 2),
                      "frac_fp32_vector": round(flops / dt / 1e12 / VEC, 4)}))


def sliding_scan():
    """time against the lag count and the code count: which part of k_sliding_dot's time is the FMA stream?"""
    nobs = 400_000
    for ncodes, nlag in ((24, 4), (24, 8), (96, 4), (96, 8), (24, 16), (24, 28), (6, 28), (96, 28)):
        n = nobs * ncodes + 64
        x = torch.randint(-3000, 3000, (n, 2), dtype=torch.int16, device=dev)
        rep = (torch.randint(0, 2, (nobs,), device=dev).float() * 2 - 1).contiguous()
        out = torch.empty((ncodes, 2 * nlag + 1, 2), dtype=torch.float64, device=dev)
        with Correlator(lfsr=(14, 43, 10000), fs=5e6) as cor:
            dt = timed(lambda: cor.sliding_dot_dev(x.data_ptr(), n, rep.data_ptr(), nobs, ncodes, nlag, out.data_ptr(), ff=1.234e-5, scale=1.0 / 32768),
                       cor.synchronize, 50)
        flops = nobs * ncodes * (2 * nlag + 1) * 4
        byts = nobs * ncodes * 4 + nobs * 4
        print(json.dumps({"kernel": "k_sliding_dot scan", "ncodes": ncodes, "nlag": nlag, "ms": round(dt * 1e3, 4), "Gsample_s": round(nobs * ncodes / dt / 1e9, 1),
                          "GB_s": round(byts / dt / 1e9, 1), "frac_hbm": round(byts / dt / 1e9 / HBM, 4),
                          "TFLOP_s": round(flops / dt / 1e12, 2), "frac_fp32_vector": round(flops / dt / 1e12 / VEC, 4)}))


def acq_batch():
    """the acquisition sweep against the trial carriers per launch (context batch size)"""
    from amaranth_twstft_amd import acquisition as A
    n_in, fs, rc, clen, nobs = 5_000_000, 10e6, 2.5e6, 100_000, 400_000
    chips = prn.lfsr_chips(17, 9, clen)
    smp = torch.randn((2 * n_in, 2), dtype=torch.float32, device=dev)
    for mb in (8, 16, 32, 64, 128):
        a = A.Acquisition(1 - 2 * chips.astype(np.int64), rc, fs, nobs, max_batch=mb)
        a.acquire(smp.data_ptr(), 3 * nobs, 186.0, 65536.0, 256.0)
        ts = []
        for _ in range(10):
            t = time.perf_counter(); a.acquire(smp.data_ptr(), 3 * nobs, 186.0, 65536.0, 256.0); ts.append(time.perf_counter() - t)
        print(json.dumps({"kernel": "acquisition sweep, one call", "max_batch": mb, "batch": int(a.cor.info.batch), "ms_median_of_10_warm": round(float(np.median(ts)) * 1e3, 3)}))
        a.close()


def acq():
    from amaranth_twstft_amd import acquisition as A
    n_in, fs, rc, clen = 5_000_000, 10e6, 2.5e6, 100_000
    nobs = 400_000
    chips = prn.lfsr_chips(17, 9, clen)
    iq = torch.randint(-2000, 2000, (n_in, 4), dtype=torch.int16, device=dev)
    smp = torch.empty((2 * n_in, 2), dtype=torch.float32, device=dev)
    interp = A.Interpolator(n_in)
    dt = timed(lambda: interp(iq.data_ptr(), smp.data_ptr(), 2, 0), interp.cor.synchronize, 10)
    byts = n_in * (4 + 4 + 8 + 8 + 8 + 16 + 16 + 16)          # sums + column pass in/out + row in, 2 phases out + in, map out
    print(json.dumps({"kernel": "interpolation chain (short2double)", "case": "5e6 -> 1e7 samples, one channel", "ms": round(dt * 1e3, 3),
                      "input_Gsample_s": round(n_in / dt / 1e9, 2), "algorithmic_bytes": byts, "GB_s": round(byts / dt / 1e9, 1), "frac_hbm": round(byts / dt / 1e9 / HBM, 4)}))
    a = A.Acquisition(1 - 2 * chips.astype(np.int64), rc, fs, nobs)
    nb = 513 + 24
    byts = nb * a.nfft * (8 + 8 + 8 + 8 + 8)                    # per trial carrier: samples in, A out/in, Bz out/in
    for name, fn in (("one call: twx_acquire_cdev (bookkeeping between rounds on the device, one synchronisation)", a.acquire),
                     ("host-driven rounds: twx_caf_freqs_cdev per round (9 synchronisations + D2H)", a.acquire_host_loop)):
        fn(smp.data_ptr(), 3 * nobs, 186.0, 65536.0, 256.0)                                      # cold call (allocations)
        times = []
        for _ in range(10):                                                                        # warm x 10
            t = time.perf_counter()
            fc, pk, pt = fn(smp.data_ptr(), 3 * nobs, 186.0, 65536.0, 256.0)
            times.append(time.perf_counter() - t)
        dt = float(np.median(times))
        print(json.dumps({"kernel": "acquisition sweep (rxcomplex.cpp:534-567)", "variant": name,
                          "case": f"{nb} trial carriers x 2^20 samples, sdr.param range/step", "ms_median_of_10_warm": round(dt * 1e3, 3),
                          "ms_min": round(min(times) * 1e3, 3), "ms_max": round(max(times) * 1e3, 3),
                          "carriers_per_s": round(nb / dt, 0), "algorithmic_bytes": byts, "GB_s": round(byts / dt / 1e9, 1), "frac_hbm": round(byts / dt / 1e9 / HBM, 4)}))
    a.close(); interp.close()


if __name__ == "__main__":
    L.load()
    for w in (sys.argv[1:] or ["fir", "sliding", "acq"]):
        globals()[w]()
