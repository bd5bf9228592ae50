"""Diagnostic: the whole correlation map of one window (k_col_fwd<MIX>, k_rowd<MID>, k_col_inv with output) beside a matrix-core FIR
on another stream against the same map alone: where do they differ?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator
lib = L.load()
dev = torch.device("cuda", 0)
N = 5_000_000; dec = 14; FS = 5e6
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
chips = prn.lfsr_chips(22, 3, 2_500_000)
g = torch.Generator(device=dev); g.manual_seed(1)
cap = (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16)
win = (torch.randn((N, 2), device=dev, generator=g) * 4000).to(torch.int16)
torch.cuda.synchronize()
with Correlator(chips, fs=FS, Nint=1) as c, Correlator(chips, fs=FS, Nint=1) as b1:
    out = torch.zeros((N, 2), dtype=torch.int16, device=dev)
    ref = torch.zeros((3 * N, 2), dtype=torch.float32, device=dev)
    L.check(lib.twx_xcorr_map_dev(c._h, win.data_ptr(), 1, 0, 1234.5, ref.data_ptr()), c._h); c.synchronize()
    maps = [torch.zeros((3 * N, 2), dtype=torch.float32, device=dev) for _ in range(4)]
    for i in range(4):
        for _ in range(2):
            b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out.data_ptr())
        L.check(lib.twx_xcorr_map_dev(c._h, win.data_ptr(), 1, 0, 1234.5, maps[i].data_ptr()), c._h)
    c.synchronize(); b1.synchronize()
    for i in range(4):
        d = (maps[i] != ref).any(dim=1)
        nbad = int(d.sum())
        print("map", i, "elements that differ:", nbad, "of", 3 * N)
        if nbad:
            idx = torch.nonzero(d).flatten().cpu().numpy()
            # natural interleaved order: m = 3 n + rho ; n = n1 * N2 + n2 (N1 = 625 columns of length..., N2 = 8000)
            n = idx // 3; rho = idx % 3
            print("  phases", np.unique(rho), " n range", n.min(), n.max(), " n2 = n % 8000:", np.unique(n % 8000)[:20], "count", np.unique(n % 8000).size,
                  " n1 = n // 8000:", np.unique(n // 8000)[:20], "count", np.unique(n // 8000).size)
            dm = (maps[i] - ref).cpu().numpy().astype(np.float64)
            dz = (dm[:, 0] + 1j * dm[:, 1])                                        # all three phases interleaved: the x3 grid
            D = np.fft.fft(dz)
            mag = np.abs(D); thr = mag.max() * 1e-3
            kb = np.nonzero(mag > thr)[0]
            kk = np.where(kb < 3 * N // 2, kb, kb - 3 * N)                            # signed bin of the 3N-point spectrum
            k = kk % N                                                              # the N-point bin it came from (zero-padded spectrum)
            print("  spectrum of the difference: bins above 1e-3 of its max:", kb.size, " k1 = k % 625:", np.unique(k % 625)[:30], "count", np.unique(k % 625).size,
                  " k2 = k // 625 range", (k // 625).min(), (k // 625).max(), "count", np.unique(k // 625).size)
            rm = ref.cpu().numpy().astype(np.float64); Rz = np.fft.fft(rm[:, 0] + 1j * rm[:, 1])
            Mz = Rz + D
            for k1 in np.unique(k % 625)[:4]:
                kk2 = np.arange(0, 4000)                                           # positive-frequency half of the row: bins k1 + 625 k2 of the 3N-point spectrum
                bins = k1 + 625 * kk2
                r = Rz[bins]; m = Mz[bins]
                ratio = m / np.where(np.abs(r) > 0, r, 1)
                good = np.abs(r) > np.abs(r).max() * 1e-3
                badk2 = kk2[np.abs(m - r) > 0.05 * np.abs(r).mean()]
                print("   row k1 =", k1, "bins k2 off by > 5 % of the row's mean:", badk2.size, " q0 = k2 % 20:", np.unique(badk2 % 20), " q1 = (k2 // 20) % 20:", np.unique((badk2 // 20) % 20),
                      " q2 = k2 // 400:", np.unique(badk2 // 400))
                print("   row k1 =", k1, ": |wrong|/|right| median %.4f  min %.4f max %.4f;  phase(wrong/right) median %.4f rad, spread %.4f;  corr(wrong,right) = %.4f" % (
                    np.median(np.abs(ratio[good])), np.abs(ratio[good]).min(), np.abs(ratio[good]).max(), np.median(np.angle(ratio[good])), np.std(np.angle(ratio[good])),
                    np.abs(np.vdot(r[good], m[good])) / np.sqrt(np.vdot(r[good], r[good]).real * np.vdot(m[good], m[good]).real)))
            rel = ((maps[i] - ref).abs().max() / ref.abs().max()).item()
            print("  max |diff| / max |ref| =", rel)
