#!/usr/bin/env python3
"""One-window calls (MEX form A, the receiver's per-second loop): what a call costs when the caller waits for every record
(call + twx_synchronize) against calls enqueued back to back, on one pipeline slot — i.e. how much of the synchronous latency is
host launch work rather than kernel time.    python tools/one_window_latency.py [calls]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import _lib as L, prn, synth  # noqa: E402
from amaranth_twstft_amd.correlator import Correlator, band_godual  # noqa: E402

N, NCHIPS, FS = 5_000_000, 2_500_000, 5e6
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = L.load()
dev = torch.device("cuda", 0)
chips = prn.lfsr_chips(22, 3, NCHIPS)
cdev = torch.from_numpy(chips).to(dev)
iq = torch.empty((N, 2), dtype=torch.int16, device=dev)
p = synth.SynthParams(delay_q8=1234567 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=1, amp=300, noise_gain=synth.noise_gain_for_sigma(500.0), seed=5)
params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
L.check(lib.twx_synth_capture_dev(iq.data_ptr(), N, 0, cdev.data_ptr(), NCHIPS, 2, 1, params.ctypes.data_as(C.c_void_p), None))
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
res = torch.zeros(C.sizeof(L.twx_result), dtype=torch.uint8, device=dev)
out = {}
for label, env in (("direct launches", {"TWX_GRAPH": "0"}), ("hipGraph", {"TWX_GRAPH": "1"})):
    os.environ.pop("TWX_GRAPH", None)
    os.environ.update(env)
    with Correlator(chips, fs=FS, Nint=1) as cor:
        def call():
            L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), 1, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
        for _ in range(5):
            call()
        cor.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            call(); cor.synchronize()
        t_sync = (time.perf_counter() - t0) / calls
        t0 = time.perf_counter()
        for _ in range(calls):
            call()
        t_enq = (time.perf_counter() - t0) / calls
        cor.synchronize()
        t_async = (time.perf_counter() - t0) / calls
        r = L.twx_result.from_buffer_copy(res.cpu().numpy().tobytes())
        out[label] = {"us_per_call_waiting_for_each": round(t_sync * 1e6, 1), "us_per_call_back_to_back": round(t_async * 1e6, 1),
                      "us_host_enqueue_per_call": round(t_enq * 1e6, 1), "indice0": int(r.indice0)}
print(json.dumps(out))
