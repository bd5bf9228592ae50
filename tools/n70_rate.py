import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from amaranth_twstft_amd.correlator import Correlator, band_godual
from amaranth_twstft_amd import synth
from tests.helpers import chips_for
from tests.test_gpu_configs import _synth_dev
dev = torch.device("cuda", 0)
fs, sps, n = 70e6, 28, 70_000_000
chips = chips_for(22, 57, 2_500_000)
p = synth.SynthParams(delay_q8=18_364_717 * 256, fstep=synth.fstep_for_df(3.25, fs), phi0=99, amp=2500, noise_gain=synth.noise_gain_for_sigma(2500.0), seed=401)
nw = 4
wide = torch.empty((nw * n, 2), dtype=torch.int16, device=dev)
_synth_dev(wide, nw * n, torch.from_numpy(chips).to(dev), 2_500_000, sps, [p])
torch.cuda.synchronize()
band = band_godual(fs, n)
with Correlator(chips, fs=fs, sps=sps, Nint=1, profile=True) as cor:
    print(cor.info.n1, cor.info.n2, cor.info.batch)
    cor.process_dev(wide.data_ptr(), nw, band=band)
    t = time.time(); g = cor.process_dev(wide.data_ptr(), nw, band=band); dt = time.time() - t
    print("windows", nw, "s/window", dt / nw, "Gsample/s", nw * n / dt / 1e9, [x.indice for x in g])
    try:
        print(cor.profile())
    except Exception as e:
        print("no profile", e)
import json
with Correlator(chips, fs=fs, sps=sps, Nint=1, profile=True) as cor:
    cor.process_dev(wide.data_ptr(), nw, band=band)
    cor.profile(reset=True)
    cor.process_dev(wide.data_ptr(), nw, band=band)
    prof = cor.profile()
    tot = sum(v["ms_total"] for v in prof.values())
    algo = {"k_sums": 4, "k_col_fwd_square": 12, "k_row_band": 8, "k_df_tables": 0, "k_col_fwd_mix": 12, "k_row_mid": 32, "k_col_inv": 24, "k_peak": 0}
    dom = max(prof, key=lambda k: prof[k]["ms_total"])
    ms = prof[dom]["ms_total"] / prof[dom]["launches"]
    spl = prof[dom]["units"] / prof[dom]["launches"]
    byts = algo[dom] * spl + (8 * n if dom == "k_row_mid" else 0)
    ach = byts / (ms * 1e-3) / 1e9
    print(json.dumps({"workload": "native 70 Msps x 1 s window (N = 7e7 = 7000 x 10000, W = 2), fp32, one pipeline slot (profiled context)", "value": round(nw * n / tot / 1e3, 1),
                      "unit": "Msamples/s", "kernels_ms": {k: round(v["ms_total"] / v["launches"], 4) for k, v in prof.items()},
                      "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "avg_ms": round(ms, 4),
                                   "algorithmic_bytes_per_launch": int(byts)},
                      "chain_GBs_algorithmic": round(92 * nw * n / tot / 1e6, 1)}))
