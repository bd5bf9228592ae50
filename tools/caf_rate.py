"""GPU box: time twx_caf_bins_dev on BASELINE.json configs[2] (N = 5e6, +-5 kHz at 1 Hz = 10 001 bins) in the DIF/DIT form
(k_rowd_caf) and, with TWX_CAF_STOCKHAM=1, in the Stockham form (k_row_caf).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import synth  # noqa: E402
from amaranth_twstft_amd.correlator import Correlator  # noqa: E402
from tests.helpers import chips_for  # noqa: E402
from tests.test_gpu_configs import _synth_dev  # noqa: E402

N, NCHIPS, FS = 5_000_000, 2_500_000, 5e6
dev = torch.device("cuda", 0)
chips = chips_for(22, 3, NCHIPS)
p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=0, amp=200,
                      noise_gain=synth.noise_gain_for_sigma(400.0), seed=7)
iq = torch.empty((N, 2), dtype=torch.int16, device=dev)
_synth_dev(iq, N, torch.from_numpy(chips).to(dev), NCHIPS, 2, [p])
torch.cuda.synchronize()
with Correlator(chips, fs=FS, Nint=0) as cor:
    cor.caf_bins_dev(iq.data_ptr(), -200, 200)
    ts = []
    for _ in range(3):
        t = time.time()
        pk, lag = cor.caf_bins_dev(iq.data_ptr(), -5000, 5000)
        ts.append(time.time() - t)
    b = int(np.argmax(pk))
print(json.dumps({"form": "stockham" if os.environ.get("TWX_CAF_STOCKHAM") else "dif/dit", "bins": 10001, "s_per_window": min(ts),
                  "us_per_bin": min(ts) / 10001 * 1e6, "peak_bin": b - 5000, "peak_lag": int(lag[b])}))
