"""Diagnostic: FIR and the chain back to back on ONE context's stream, with and without a wait in between."""
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
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df)
for mode in ("wait between", "back to back", "back to back"):
    with Correlator(chips, fs=FS, Nint=1) as c:
        nar = torch.zeros((N, 2), dtype=torch.int16, device=dev)
        res = torch.zeros((6, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
        for i in range(6):
            c.fir_decimate_dev(caps[i % 2].data_ptr(), n_in, taps, dec, out_i16_dev=nar.data_ptr())
            if mode == "wait between":
                c.synchronize()
            L.check(lib.twx_process_windows_dev(c._h, nar.data_ptr(), 1, 1, 0, C.byref(band), None, res[i].data_ptr()), c._h)
        c.synchronize()
        recs = [key(L.twx_result.from_buffer_copy(res[i].cpu().numpy().tobytes())) for i in range(6)]
        if mode == "wait between":
            base = recs
        else:
            print(mode, [i for i in range(6) if recs[i] != base[i]])
print("done")
