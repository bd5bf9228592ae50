#!/usr/bin/env python3
"""PCIe- and file-inclusive rate of the file-in/results-out path; never bench.py's `value`.
    python tools/io_rate.py [n_windows] [io_threads ...]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from amaranth_twstft_amd import prn
from amaranth_twstft_amd.correlator import Correlator, band_godual
if os.environ.get("TWX_IO_CPUS"):                      # e.g. "0-63": writer and readers on the CPUs of one NUMA node (profiles/r05_io_rate.txt)
    lo, hi = os.environ["TWX_IO_CPUS"].split("-")
    os.sched_setaffinity(0, range(int(lo), int(hi) + 1))
NCH = 2500000; N = 2 * NCH; NW = int(sys.argv[1]) if len(sys.argv) > 1 else 48
threads = [int(a) for a in sys.argv[2:]] or [4]
chips = prn.lfsr_chips(22, 3, NCH)
rng = np.random.default_rng(1)
code = np.repeat(chips.astype(np.int16), 2) * 2 - 1
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    path = os.path.join(td, "1670000000.bin")
    block = []
    for w in range(8):                                          # 8 distinct windows, repeated
        x = (200 * np.roll(code, 1311765 - w)).astype(np.int16)
        block.append((np.stack([x, np.zeros_like(x)], axis=1) + rng.integers(-400, 400, (N, 2), dtype=np.int16)).astype(np.int16))
    block = np.concatenate(block)
    with open(path, "wb") as f:
        for _ in range((NW + 7) // 8):
            block.tofile(f)
    band = band_godual(5e6, N)
    for t in threads:
        os.environ["TWX_IO_THREADS"] = str(t)
        with Correlator(chips, fs=5e6, Nint=1) as cor:
            cor.process_file(path, n_channels=1, channel=0, band=band, max_windows=8)      # warm
            t0 = time.perf_counter()
            res = cor.process_file(path, n_channels=1, channel=0, band=band, max_windows=NW)
            dt = time.perf_counter() - t0
        ok = all(r.indice == 3 * (1311765 - (w % 8)) for w, r in enumerate(res)) and len(res) == NW
        print(f"process_file io_threads={t}: {NW} windows in {dt*1e3:.1f} ms = {NW*N/dt/1e6:.0f} Msample/s ({NW*N*4/dt/1e9:.1f} GB/s of int16), lags ok={ok}")
    with Correlator(chips, fs=5e6, Nint=1) as cor:
        raw = np.fromfile(path, dtype=np.int16, count=min(NW, 48) * N * 2)
        cor.process(raw[:8 * N * 2], n_channels=1, channel=0, band=band)
        t1 = time.perf_counter()
        res2 = cor.process(raw, n_channels=1, channel=0, band=band)
        dt2 = time.perf_counter() - t1
    print(f"process (host buffer, pageable, {len(res2)} windows): {dt2*1e3:.1f} ms = {len(res2)*N/dt2/1e6:.0f} Msample/s")
    # what the link itself gives: one 160-MB pinned buffer copied to the device, 20 times back to back
    import torch
    h = torch.empty(160 << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(160 << 20, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(20):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt3 = time.perf_counter() - t2
    print(f"pinned H2D, 20 x 160 MB: {20 * (160 << 20) / dt3 / 1e9:.1f} GB/s = {20 * (160 << 20) / 4 / dt3 / 1e6:.0f} Msample/s of int16 IQ")
