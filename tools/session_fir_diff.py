import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator
from amaranth_twstft_amd.wideband import WidebandSession, godual_plan
dev = torch.device("cuda", 0)
N = 5_000_000; dec = 14; FS = 5e6
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
codes = {"OP": prn.lfsr_chips(22, 57, 2_500_000), "LTFB": prn.lfsr_chips(22, 3, 2_500_000)}
g = torch.Generator(device=dev); g.manual_seed(1)
sets = [{st: (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16) for st in codes} for _ in range(2)]
torch.cuda.synchronize()
ref = []
with Correlator(codes["OP"], fs=FS, Nint=1) as c:
    for cs in sets:
        out = {}
        for st in codes:
            o = torch.zeros((N, 2), dtype=torch.int16, device=dev)
            c.fir_decimate_dev(cs[st].data_ptr(), n_in, taps, dec, out_i16_dev=o.data_ptr()); c.synchronize()
            out[st] = o
        ref.append(out)
plan = godual_plan(("OP", "LTFB"), FS, N)
with WidebandSession(codes, taps, dec, fs=FS, windows=1, plan=plan) as sess:
    for i in range(5):
        sess.submit({st: t.data_ptr() for st, t in sets[i % 2].items()})
        sess.synchronize()
        for st in codes:
            d = sess.decimated(st, i)
            bad = torch.nonzero((d != ref[i % 2][st]).any(dim=1)).flatten().cpu().numpy()
            if bad.size:
                print("step", i, st, "differs at", bad.size, "outputs; first", bad[:8], "last", bad[-4:], "trips", np.unique(bad // 256)[:10])
    print("synchronised steps: done")
key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df)
for mode in ("one at a time", "pipelined", "pipelined"):
    with WidebandSession(codes, taps, dec, fs=FS, windows=1, plan=plan, depth=6) as sess:
        got = {}
        for i in range(6):
            sess.submit({st: t.data_ptr() for st, t in sets[i % 2].items()})
            if mode == "one at a time":
                sess.synchronize()
        sess.synchronize()
        for i in range(6):
            got[i] = {k: key(v[0]) for k, v in sess.fetch(i).items()}
        for i in range(6):
            for st in codes:
                d = sess.decimated(st, i)
                bad = torch.nonzero((d != ref[i % 2][st]).any(dim=1)).flatten().cpu().numpy()
                if bad.size:
                    print(mode, "step", i, st, "decimated differs at", bad.size, "outputs; first", bad[:8], "trips", np.unique(bad // 256)[:10])
        if mode == "one at a time":
            base = got
        else:
            for i in range(6):
                for k in plan:
                    if got[i][k] != base[i % 2][k]:
                        print(mode, "step", i, k, "record differs:", got[i][k], base[i % 2][k])
print("done")
