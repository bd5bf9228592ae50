#!/usr/bin/env python3
"""Per-kernel times of processing(d,k) at another window length or precision (python tools/prof_n.py bitlen taps nchips [f32|f64]), with a
roofline object for the dominant kernel (algorithmic bytes as in DESIGN.md section 4, complex element = 8 B fp32 / 16 B fp64)."""
import json
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from amaranth_twstft_amd import _lib as L, prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual
FS = 5e6
bitlen, taps, nchips = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (17, 9, 100000)))
precision = sys.argv[4] if len(sys.argv) > 4 else "f32"
n = 2 * nchips
lib = L.load()
chips = prn.lfsr_chips(bitlen, taps, nchips)
nwin = max(16, min(4096, int(160e6 // n)))
p = synth.SynthParams(delay_q8=(n // 3) * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=1, amp=300, noise_gain=synth.noise_gain_for_sigma(500.0), seed=5)
raw = torch.from_numpy(synth.synth_channel(n * nwin, chips, 2, p).reshape(-1)).cuda()
res = torch.zeros((nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device="cuda")
band = L.twx_band(*band_godual(FS, n))
with Correlator(chips, fs=FS, Nint=1, profile=True, precision=precision) as cor:
    def run():
        L.check(lib.twx_process_windows_dev(cor._h, raw.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
        L.check(lib.twx_synchronize(cor._h), cor._h)
    run(); cor.profile(reset=True); run(); run()
    prof = cor.profile()
    tot = sum(v["ms_total"] for v in prof.values())
    print(f"{precision} N={n} N1={cor.info.n1} N2={cor.info.n2} B={cor.info.batch}: {2 * nwin * n / tot / 1e6:.1f} Gsample/s (one slot)")
    for k, v in prof.items():
        print(f"  {k:18s} {v['ms_total'] / v['launches'] * 1e3:8.1f} us/launch  {v['ms_total'] / tot * 100:5.1f} %  {v['units'] / v['launches'] / 1e6:.2f} M samples/launch")
    E = 16 if precision == "f64" else 8                    # bytes per complex element of the intermediates
    R = 3
    algo = {"k_sums": lambda S: 4 * S, "k_col_fwd_square": lambda S: (4 + E) * S, "k_row_band": lambda S: E * S, "k_df_tables": lambda S: 0,
            "k_col_fwd_mix": lambda S: (4 + E) * S, "k_row_mid": lambda S: (1 + R) * E * S + E * n, "k_col_inv": lambda S: R * E * S, "k_peak": lambda S: 0}
    kern = {k: (v["ms_total"] / v["launches"], v["units"] / v["launches"]) for k, v in prof.items()}
    dom = max(kern, key=lambda k: prof[k]["ms_total"])
    ach = algo[dom](kern[dom][1]) / (kern[dom][0] * 1e-3) / 1e9
    chain = sum(algo[k](kern[k][1]) for k in kern) / kern[dom][1] * (2 * nwin * n / tot / 1e3) / 1e9
    print(json.dumps({"workload": f"processing(d,k), {precision}, N = {n}, one pipeline slot, HBM-resident", "value": round(2 * nwin * n / tot / 1e3, 1), "unit": "Msamples/s",
                      "bytes_per_sample": sum(algo[k](1e6) for k in kern) / 1e6,
                      "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                                   "avg_ms": round(kern[dom][0], 4), "algorithmic_bytes_per_launch": int(algo[dom](kern[dom][1]))},
                      "chain_GBs_algorithmic": round(chain * 1e6, 1), "chain_frac": round(chain * 1e6 / 8000.0, 4)}))
