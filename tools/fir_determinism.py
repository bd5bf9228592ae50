import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator
lib = L.load()
dev = torch.device("cuda", 0)
N = 5_000_000; dec = 14
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16)
torch.cuda.synchronize()
chips = prn.lfsr_chips(22, 3, 2_500_000)
outs = {}
for label, env in (("mfma_a", "1"), ("mfma_b", "1"), ("vec", "0")):
    os.environ["TWX_FIR_MFMA"] = env
    with Correlator(chips, fs=5e6, Nint=1) as c:
        o16 = torch.zeros((N, 2), dtype=torch.int16, device=dev); of = torch.zeros((N, 2), dtype=torch.float32, device=dev)
        for rep in range(3):
            assert c.fir_decimate_dev(x.data_ptr(), n_in, taps, dec, out_i16_dev=o16.data_ptr(), out_f32_dev=of.data_ptr()) == N
        c.synchronize()
        outs[label] = (o16.clone(), of.clone())
a, b, v = outs["mfma_a"], outs["mfma_b"], outs["vec"]
print("mfma run-to-run equal:", bool((a[0] == b[0]).all()), bool((a[1] == b[1]).all()))
d16 = (a[0].to(torch.int32) - v[0].to(torch.int32)).abs().amax(dim=1)
bad = torch.nonzero(d16 > 1).flatten()
print("int16 |mfma - vec| max", int(d16.max()), "count > 1:", int(bad.numel()))
if bad.numel():
    bi = bad.cpu().numpy()
    print("first bad", bi[:20], "trips", np.unique(bi // 256)[:20], "pos in trip", np.unique(bi % 256)[:40])
df = (a[1] - v[1]).abs().max().item(); print("float max diff", df, "ref max", v[1].abs().max().item())
