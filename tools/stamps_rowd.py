#!/usr/bin/env python3
"""Diagnostic: where a workgroup of k_rowd<MID> spends its cycles — s_memtime stamps of one lane per wave at the segment boundaries
(a -DTWX_STAMPS variant: `tools/variants.sh stamps -DTWX_STAMPS`, then TWX_LIB=amaranth_twstft_amd/variants/lib_stamps.so).
Not a timing: the stamps themselves cost cycles (MI355X_MICROARCH.md: about +11 %)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TWX_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amaranth_twstft_amd", "variants", "lib_stamps.so"))
from amaranth_twstft_amd import _lib as L, prn
from amaranth_twstft_amd.correlator import Correlator
import torch
lib = L.load()
NCH = 2500000; N = 2 * NCH; B = 8
chips = prn.lfsr_chips(22, 3, NCH)
os.environ["TWX_STREAMS"] = "1"
cor = Correlator(chips, fs=5e6, Nint=1, max_batch=B)
iq = torch.randint(-500, 500, (B, N, 2), dtype=torch.int16, device="cuda")
res = torch.zeros((B, 240), dtype=torch.uint8, device="cuda")
df = np.full(B, 1780.75)
for _ in range(3):
    L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), B, 1, 0, None, df.ctypes.data_as(C.c_void_p), res.data_ptr()), cor._h)
L.check(lib.twx_synchronize(cor._h))
rows = B * 625
nwg = int(os.environ.get("TWX_ROW_PF", "512"))       # resident workgroups (two per CU); the stamps of a workgroup's LAST row remain
nwg = rows if nwg <= 0 else min(rows, nwg)
st = np.zeros(rows * 7 * 32, dtype=np.uint64)
L.check(lib.twx_debug_stamps(cor._h, st.ctypes.data_as(C.c_void_p), st.size))
st = st.reshape(rows, 7, 32).astype(np.int64)[:nwg]
lab = ["row top: take the row loaded ahead, first butterfly", "barrier 1", "wait row loads + fwd stage 0 (bfly, twiddle, LDS wr)", "barrier 2 (all-to-all)", "fwd stage 1 (table fold, rd, bfly, wr)",
       "wave sync + fwd stage 2 + product + iA_pre"]
for r in range(3):
    lab += [f"rho{r}: barrier (top)", f"rho{r}: iA store + wave sync", f"rho{r}: stage B (rd, bfly, twiddle, wr)", f"rho{r}: barrier (all-to-all)",
            f"rho{r}: stage C (rd, bfly, twiddle, global st)", f"rho{r}: next phase ramp + iA_pre"]
nseg = len(lab)
d = np.diff(st[:, :, :nseg + 1], axis=2)
ok = (st[:, :, :nseg + 1] > 0).all(axis=2)
med = np.median(d[ok], axis=0)
tot = med.sum()
for i in range(nseg):
    print(f"{lab[i]:56s} {med[i]:8.0f} cyc {100 * med[i] / tot:5.1f} %")
life = st[:, :, nseg] - st[:, :, 0]
print(f"sum of medians {tot:.0f} cycles; one row, median over waves {np.median(life[ok]):.0f} cycles (s_memtime tick = shader cycle)")
whole = (st[:, :, 31] - st[:, :, 30])[ok]
nrows = np.array([len(range(g, rows, nwg)) for g in range(nwg)])[:, None] * np.ones((1, 7), dtype=np.int64)
print(f"workgroup lifetime (first to last stamp): median {np.median(whole):.0f}, max {whole.max():.0f} cycles; rows per workgroup {nrows.min()}..{nrows.max()}; "
      f"lifetime / rows: median {np.median(whole / nrows[ok]):.0f} cycles per row")
t0 = st[:, :, 30][ok].min(); t1 = st[:, :, 31][ok].max()
print(f"first start to last end over all workgroups: {t1 - t0} cycles; start spread {st[:, :, 30][ok].max() - t0}, end spread {t1 - st[:, :, 31][ok].min()}")
grp = {"forward (stage 0-2 + product)": range(0, 6), "barriers": [1, 3] + [6 + 6 * r for r in range(3)] + [9 + 6 * r for r in range(3)],
       "stage B": [8 + 6 * r for r in range(3)], "stage C + stores": [10 + 6 * r for r in range(3)], "ramp + stage A": [7 + 6 * r for r in range(3)] + [11 + 6 * r for r in range(3)]}
for k, idx in grp.items():
    print(f"  {k:32s} {100 * sum(med[i] for i in idx) / tot:5.1f} %")
