#!/usr/bin/env python3
"""TWX_OPT_SELFCHECK beside the fault it exists for (python tools/selfcheck_soak.py [calls]).

One correlator context runs a 1-s window with the Parseval self-check on while another context runs the FIR front end on its own
stream, nothing synchronised in between: vector form, matrix-core form as the library runs it (fenced: never co-resident) and the
matrix-core form UNFENCED (TWX_FIR_MFMA_UNFENCED=1, diagnostic) — the configuration of profiles/r05_fir_mfma.txt in which rows of the
correlation's k_rowd go wrong.  Per mode: records that differ from the run alone, records flagged, rows flagged, largest relative
Parseval deviation.  Every wrong record must be flagged; nothing must be flagged where nothing is wrong."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator, band_godual

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lib = L.load()
dev = torch.device("cuda", 0)
N, dec, FS = 5_000_000, 14, 5e6
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
chips = prn.lfsr_chips(22, 3, 2_500_000)
g = torch.Generator(device=dev); g.manual_seed(1)
cap = (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16)
win = [(torch.randn((N, 2), device=dev, generator=g) * 4000).to(torch.int16) for _ in range(2)]
out16 = torch.zeros((N, 2), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
RB = C.sizeof(L.twx_result)
key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df, r.SNRr)


def stats(c):
    d, n = C.c_double(), C.c_int64()
    L.check(lib.twx_selfcheck_stats(c._h, C.byref(d), C.byref(n), 1), c._h)
    return d.value, n.value


with Correlator(chips, fs=FS, Nint=1) as c, Correlator(chips, fs=FS, Nint=1) as b1:
    L.check(lib.twx_set_option(c._h, L.TWX_OPT_SELFCHECK, 1), c._h)
    chain = lambda i, r: L.check(lib.twx_process_windows_dev(c._h, win[i % 2].data_ptr(), 1, 1, 0, C.byref(band), None, r.data_ptr()), c._h)
    alone = []
    for i in range(2):
        r = torch.zeros(RB, dtype=torch.uint8, device=dev); chain(i, r); c.synchronize()
        alone.append(key(L.twx_result.from_buffer_copy(r.cpu().numpy().tobytes())))
    stats(c)
    for mode in ("alone", "vector FIR", "matrix-core FIR (fenced: as shipped)", "matrix-core FIR UNFENCED (diagnostic)"):
        L.check(lib.twx_set_option(b1._h, L.TWX_OPT_FIR_MFMA, 1 if "matrix" in mode else 0), b1._h)
        if "UNFENCED" in mode:
            os.environ["TWX_FIR_MFMA_UNFENCED"] = "1"
        res = torch.zeros((calls, RB), dtype=torch.uint8, device=dev)
        for i in range(calls):
            if mode != "alone":
                b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out16.data_ptr())
            chain(i, res[i])
        c.synchronize(); b1.synchronize(); torch.cuda.synchronize()
        os.environ.pop("TWX_FIR_MFMA_UNFENCED", None)
        host = res.cpu().numpy()
        recs = [L.twx_result.from_buffer_copy(host[i].tobytes()) for i in range(calls)]
        wrong = [i for i in range(calls) if key(recs[i]) != alone[i % 2]]
        flagged = [i for i in range(calls) if recs[i].status & L.TWX_STATUS_SELFCHECK]
        worst, rows = stats(c)
        print(json.dumps({"mode": mode, "calls": calls, "wrong_records": wrong, "flagged_records": flagged, "wrong_but_not_flagged": sorted(set(wrong) - set(flagged)),
                          "flagged_but_not_wrong": sorted(set(flagged) - set(wrong)), "rows_flagged": rows, "largest_relative_deviation": float("%.3g" % worst)}), flush=True)
