"""Diagnostic: the front end on one context's stream while two other contexts run the chain: (a) are its outputs the same as alone,
(b) does a chain that follows it on the same stream see them."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator, band_godual
lib = L.load()
dev = torch.device("cuda", 0)
N = 5_000_000; dec = 14; FS = 5e6
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
chips = prn.lfsr_chips(22, 3, 2_500_000)
g = torch.Generator(device=dev); g.manual_seed(1)
caps = [(torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16) for _ in range(2)]
noise = (torch.randn((4 * N, 2), device=dev, generator=g) * 4000).to(torch.int16)
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df)
with Correlator(chips, fs=FS, Nint=1) as c, Correlator(chips, fs=FS, Nint=1) as b1, Correlator(chips, fs=FS, Nint=1) as b2:
    alone = []
    for i in range(2):
        o = torch.zeros((N, 2), dtype=torch.int16, device=dev)
        c.fir_decimate_dev(caps[i].data_ptr(), n_in, taps, dec, out_i16_dev=o.data_ptr()); c.synchronize()
        r = torch.zeros(C.sizeof(L.twx_result), dtype=torch.uint8, device=dev)
        L.check(lib.twx_process_windows_dev(c._h, o.data_ptr(), 1, 1, 0, C.byref(band), None, r.data_ptr()), c._h); c.synchronize()
        alone.append((o, key(L.twx_result.from_buffer_copy(r.cpu().numpy().tobytes()))))
    other = [torch.zeros((N, 2), dtype=torch.int16, device=dev) for _ in range(2)]
    for MODE in ("chains", "firs", "firs+chains", "firs, chain on a fixed copy"):
        print(MODE)
        outs = [torch.zeros((N, 2), dtype=torch.int16, device=dev) for _ in range(8)]
        res = torch.zeros((8, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
        rb = torch.zeros((8, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
        for i in range(8):
            if MODE == "chains":
                L.check(lib.twx_process_windows_dev(b1._h, noise.data_ptr(), 4, 1, 0, C.byref(band), None, rb.data_ptr()), b1._h)
                L.check(lib.twx_process_windows_dev(b2._h, noise.data_ptr(), 4, 1, 0, C.byref(band), None, rb[4:].data_ptr()), b2._h)
            else:
                b1.fir_decimate_dev(caps[(i + 1) % 2].data_ptr(), n_in, taps, dec, out_i16_dev=other[i % 2].data_ptr())
                if MODE == "firs+chains":
                    L.check(lib.twx_process_windows_dev(b1._h, other[i % 2].data_ptr(), 1, 1, 0, C.byref(band), None, rb.data_ptr()), b1._h)
                    L.check(lib.twx_process_windows_dev(b2._h, noise.data_ptr(), 4, 1, 0, C.byref(band), None, rb[4:].data_ptr()), b2._h)
            c.fir_decimate_dev(caps[i % 2].data_ptr(), n_in, taps, dec, out_i16_dev=outs[i].data_ptr())
            src = alone[i % 2][0] if MODE.endswith("fixed copy") else outs[i]
            L.check(lib.twx_process_windows_dev(c._h, src.data_ptr(), 1, 1, 0, C.byref(band), None, res[i].data_ptr()), c._h)
        c.synchronize(); b1.synchronize(); b2.synchronize()
        for i in range(8):
            same = bool((outs[i] == alone[i % 2][0]).all())
            rec = key(L.twx_result.from_buffer_copy(res[i].cpu().numpy().tobytes()))
            print(i, "front end output as alone:", same, " record as alone:", rec == alone[i % 2][1])
