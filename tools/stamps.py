#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of k_row<MID> from the TWX_STAMPS build (not a timing)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TWX_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amaranth_twstft_amd", "libtwx_stamps.so"))
from amaranth_twstft_amd import _lib as L, prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual
import torch
lib = L.load()
lib.twx_debug_stamps.restype = C.c_int; lib.twx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
NCH = 2500000; N = 2 * NCH; B = 8
chips = prn.lfsr_chips(22, 3, NCH)
cor = Correlator(chips, fs=5e6, Nint=1, max_batch=B)
dev = torch.device("cuda", 0)
iq = torch.randint(-500, 500, (B, N, 2), dtype=torch.int16, device=dev)
res = torch.zeros((B, 240), dtype=torch.uint8, device=dev)
df = np.full(B, 1780.75)
for _ in range(3):
    L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), B, 1, 0, None, df.ctypes.data_as(C.c_void_p), res.data_ptr()), cor._h)
L.check(lib.twx_synchronize(cor._h))
nwg = B * 625
st = np.zeros(nwg * 7 * 32, dtype=np.uint64)
L.check(lib.twx_debug_stamps(cor._h, st.ctypes.data_as(C.c_void_p), st.size))
st = st.reshape(nwg, 7, 32).astype(np.int64)
names = ["entry→stage0 done", "barrier", "fwd stage1", "fwd stage2+prod(→loop)", "loop setup"]
d = np.diff(st[:, :, :23], axis=2)
med = np.median(d.reshape(-1, 22), axis=0)
lab = ["fwd s0 (gld+bfly+wr)", "barrier", "fwd s1 (rd..wr, 2 barriers)", "fwd s2 + product", "loop entry"]
for r in range(3):
    lab += [f"rho{r}: ramp+bfly0+wr", f"rho{r}: barrier", f"rho{r}: s1 (2 barriers)", f"rho{r}: s2 rd+bfly", f"rho{r}: twiddle+store", f"rho{r}: loop back"]
tot = med[:22].sum()
for i in range(22):
    print(f"{lab[i]:32s} {med[i]:9.0f} cyc  {100*med[i]/tot:5.1f}%")
print("total per row (median wave):", tot, "cycles;  wave0 lifetime median", np.median(st[:, 0, 22] - st[:, 0, 0]))
