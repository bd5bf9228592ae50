"""GPU box: the tracked multi-code flow (twx_tracked_file, acquisition/claudio_aligned_code_ranging_separate.m:143-205 and its `re` /
`lo` siblings) at the reference's record length — a 180-s single-channel sc16 capture (3.6 GB, generated on the device, written to
/tmp), 100 000-chip code at 5 Msps (40-ms codes, 2-s chunks), all three modes, with the per-stage breakdown of twx_tracked_timing.
    python tools/tracked_rate.py [seconds]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import _lib as L, prn, synth  # noqa: E402
from amaranth_twstft_amd.tracked import TrackedRanging  # noqa: E402
from tests.test_gpu_configs import _synth_dev  # noqa: E402

FS, NCHIPS, N = 5e6, 100_000, 200_000
seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 180
dev = torch.device("cuda", 0)
chips = prn.lfsr_chips(17, 9, NCHIPS)
chips_dev = torch.from_numpy(chips).to(dev)
path = "/tmp/twx_tracked_%ds.bin" % seconds


def write_capture(df_hz):
    """delay drifting by 5 ns/s (220616_Besancon/README.md:35) is below a sample over the record: constant delay, carrier df_hz"""
    p = synth.SynthParams(delay_q8=123_456 * 256 + 77, fstep=synth.fstep_for_df(df_hz, FS), phi0=9, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=21)
    buf = torch.empty((int(FS), 2), dtype=torch.int16, device=dev)
    with open(path, "wb") as f:
        for s in range(seconds):
            _synth_dev(buf, int(FS), chips_dev, NCHIPS, 2, [p], n0=s * int(FS))
            torch.cuda.synchronize()
            f.write(buf.cpu().numpy().tobytes())


for mode, OP, df_hz in (("ranging", 0, 12.0), ("re", 0, 50_012.0), ("lo", 0, 12.0)):
    write_capture(df_hz)
    with TrackedRanging(chips, fs=FS, Nint=1, mode=mode, OP=OP) as tr:
        for rep in range(2):                                  # first pass also pages the file in
            t = time.time()
            out = tr.run_file(path, skip_seconds=0.0)
            dt = time.time() - t
        tm = tr.timing()
    ncodes = len(out["indice1"])
    line = {"mode": mode, "record_s": seconds, "codes": ncodes, "chunks": len(out["df"]), "batches": out["batches"], "moved": len(out["moved"]),
            "kbon": out["kbon"], "s": round(dt, 4), "Gsample_s_file_inclusive": round(seconds * FS / dt / 1e9, 2),
            "x_real_time": round(seconds / dt, 1),
            "stages_ms": {k: [round(v[0] * 1e3, 2), v[1]] for k, v in tm.items()},
            "delay_samples_median": float(np.median(np.array(out["indice1"]) + np.array(out["correction1"])))}
    print(json.dumps(line))
os.unlink(path)
