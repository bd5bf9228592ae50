#!/usr/bin/env python3
"""Benchmark of the TWSTFT correlation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the ranks are ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...``
(one process per GPU, RCCL); a plain ``python bench.py --gpus N`` starts exactly that as a child process before
touching the GPU and relays its output.

One *step* = one pass of ``processing(d,k)`` (processing/Octave/godual_ranging.m:12-49: coarse carrier estimate,
NCO mix, FFT circular xcorr with x3 interpolation, peak pick, parabolic refinement, wipe-off SNR) over one
10-minute recording per GPU: ``--windows`` (default 600) 1-second windows of 5 Msps int16 IQ against the 2.5 Mchip
LFSR(22, taps 3) code (BASELINE.json configs[1]'s window, configs[3]'s recording length), already resident in HBM
(12 GB per GPU).  Windows are independent, so with N GPUs every rank processes its own recording (weak scaling) and
the per-window results are gathered with one RCCL all_gather per step.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver

FS = 5e6
BITLEN, TAPS, NCHIPS = 22, 3, 2_500_000
N = 2 * NCHIPS
HBM_PEAK_GBS = 8000.0

# compulsory (algorithmic) HBM bytes of one launch over S samples of N-sample windows, fp32
# (DESIGN.md §4): int16 IQ in = 4 B, complex fp32 = 8 B; k_row_mid reads A (8), writes 3 phases (24)
# and reads the code spectrum once per launch (8*N, shared by all windows of the launch).
ALGO_BYTES = {"k_sums": lambda S: 4 * S, "k_col_fwd_square": lambda S: 12 * S, "k_row_band": lambda S: 8 * S,
              "k_df_tables": lambda S: 0, "k_col_fwd_mix": lambda S: 12 * S, "k_row_mid": lambda S: 32 * S + 8 * N,
              "k_col_inv": lambda S: 24 * S, "k_peak": lambda S: 0}


def window_params(p: int, rank: int):
    """Per-window synthetic parameters (channel 1)."""
    from amaranth_twstft_amd import synth
    g = rank * 100000 + p
    delay = 1311765 - int(round(0.025 * g))
    df = 1780.75 + 0.05 * math.sin(2 * math.pi * g / 600)
    return synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(df, FS), phi0=(g * 2654435761) & 0xFFFFFFFF,
                             amp=200, noise_gain=synth.noise_gain_for_sigma(400.0), seed=1000 + g, stream=0), delay


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy fp64 restatement of processing(), = the reference's numpy path) on the host cores.
# Runs in a CHILD process that never touches the GPU, started before the parent initialises HIP, so that its worker
# pool can fork freely.
# ------------------------------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_gen(p):
    from amaranth_twstft_amd import synth
    sp, _ = window_params(p, 0)
    return synth.synth_channel(N, _CPU["chips"], 2, sp)


def _cpu_one(p):
    import numpy as np
    orc = _CPU["orc"]
    d = orc.deinterleave(_CPU["raw"][p % len(_CPU["raw"])], 1, 0)
    d = d - d.mean()
    r = orc.processing(d, _CPU["k"], _CPU["freq"], _CPU["temps"], _CPU["fcode"], _CPU["code"], Nint=1, fs=FS, df=_CPU["df"])
    return int(r["indice"]), float(r["correction"]), float(abs(r["xval"]))


def cpu_baseline_child(n_win: int, workload: str, max_workers: int):
    import multiprocessing as mp
    import numpy as np
    from amaranth_twstft_amd import prn
    from oracle import twstft_oracle as orc
    cores = os.cpu_count() or 1
    try:
        import psutil
        by_mem = max(1, int(psutil.virtual_memory().available // (3 << 30)))     # ~2.5 GB peak per worker (3N complex128 temporaries)
    except Exception:
        by_mem = cores
    workers = max(1, min(cores, by_mem, max_workers))
    chips = prn.lfsr_chips(BITLEN, TAPS, NCHIPS)
    _CPU.update(chips=chips, orc=orc, df=None if workload == "processing" else 1780.75)
    ctx = mp.get_context("fork")
    with ctx.Pool(min(workers, n_win)) as pool:                  # synthetic windows (untimed), same bytes as the GPU generator
        _CPU["raw"] = pool.map(_cpu_gen, range(n_win))
    t1 = time.perf_counter()
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    freq = orc.freq_axis(FS, N)
    _CPU.update(code=code, fcode=fcode, freq=freq, k=orc.band_godual(freq), temps=np.arange(N) / FS)
    t2 = time.perf_counter()
    full = [_cpu_one(p) for p in range(n_win)]                   # (i) one core, comparable with the README timings
    t3 = time.perf_counter()
    indices = [f[0] for f in full]
    out = {"single_core": {"value": round(n_win * N / (t3 - t2) / 1e6, 4), "cores": 1, "windows": n_win, "seconds": round(t3 - t2, 2)},
           "setup_seconds": round(t2 - t1, 2), "indices": indices, "corrections": [f[1] for f in full], "peak_mags": [f[2] for f in full],
           "host_cores": cores}
    if workers > 1:                                              # (ii) one window per worker process, `workers` of them side by side (the variant is named by that count)
        with ctx.Pool(workers) as pool:
            pool.map(_cpu_one, range(workers))                   # warm-up: page in the inherited arrays
            t4 = time.perf_counter()
            got = pool.map(_cpu_one, range(2 * workers), chunksize=1)
            t5 = time.perf_counter()
        ok = all(got[i][0] == indices[i % n_win] for i in range(len(got)))
        out["workers_%d" % workers] = {"value": round(2 * workers * N / (t5 - t4) / 1e6, 4), "cores": workers, "windows": 2 * workers,
                                       "seconds": round(t5 - t4, 2), "consistent": bool(ok),
                                       "note": "%d worker processes on a host of %d cores, one window each at a time — NOT all cores: a worker holds 2.5 GB "
                                               "(3N complex128 temporaries) and the memory limit of the box is not visible from inside it, so the count is "
                                               "capped (--cpu-max-workers); from 1 to 64 workers the rate grows 7x, i.e. numpy's fp64 processing() is already "
                                               "bound by the host's memory system there" % (workers, cores)}
    try:                                                         # (iii) multi-threaded FFT (scipy.fft workers=-1)
        orc.use_fft_backend("scipy", workers=-1)
        _CPU["fcode"] = orc.make_fcode(code)
        t6 = time.perf_counter()
        got = [_cpu_one(p) for p in range(n_win)]
        t7 = time.perf_counter()
        out["scipy_fft_workers_all"] = {"value": round(n_win * N / (t7 - t6) / 1e6, 4), "cores": cores, "windows": n_win,
                                        "seconds": round(t7 - t6, 2), "consistent": [g[0] for g in got] == indices}
    except Exception as e:  # pragma: no cover
        out["scipy_fft_workers_all"] = {"error": repr(e)}
    finally:
        orc.use_fft_backend("numpy")
    print("CPU_BASELINE_JSON " + json.dumps(out), flush=True)


def run_cpu_baseline(n_win: int, workload: str, max_workers: int):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-windows", str(n_win),
           "--workload", workload, "--cpu-max-workers", str(max_workers)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            return json.loads(line[len("CPU_BASELINE_JSON "):])
    return {"error": (r.stdout[-500:] + r.stderr[-1500:])}


def measure_pmc_traffic(kernel_short, child_args=("--steps", "1", "--warmup", "0", "--windows", "8", "--no-cpu-baseline", "--no-roofline", "--no-caf"),
                        what="1 step, 8 windows = one launch of the kernel"):
    """HBM bytes per launch of a kernel (or, for a tuple of names, of each of them) from the PMC counters, measured by THIS
    invocation: two child runs of a small bench under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no
    trace domains), read = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md, section HBM), write = WRITE_SIZE;
    counters in KiB.  Runs before this process touches the GPU.  Returns (bytes or None, note) — bytes a dict for a tuple."""
    many = not isinstance(kernel_short, str)
    names = tuple(kernel_short) if many else (kernel_short,)
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ):
        return None, "this run is itself under a profiler"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from summarize_prof import short
    except Exception as e:  # pragma: no cover
        return None, "tools/summarize_prof.py: %r" % (e,)
    tmp = tempfile.mkdtemp(prefix="twx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "--", sys.executable,
                   os.path.abspath(__file__)] + list(child_args)
            # its own process group: on a timeout the whole group goes (rocprofv3 AND the bench process under it, which would
            # otherwise keep running on the GPU through the parent's timed region), and is waited for before the next GPU user
            proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, "rocprofv3 --pmc %s timed out after 240 s (its process group was killed and waited for)" % counter
            r = subprocess.CompletedProcess(cmd, proc.returncode, so, se)
            files = glob.glob(os.path.join(tmp, counter, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %s): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-300:])
            best = {n: 0.0 for n in names}
            for path in files:
                for row in csv.DictReader(open(path)):
                    if row["Counter_Name"] != counter:
                        continue
                    sn = short(row["Kernel_Name"])
                    for n in names:
                        if sn.startswith(n):
                            best[n] = max(best[n], float(row["Counter_Value"]))  # per launch: the fullest launch of the run
            if min(best.values()) <= 0:
                return None, "no %s rows for %s" % (counter, [n for n in names if best[n] <= 0])
            vals[counter] = best
    except Exception as e:
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    byts = {n: int(2 * vals["FETCH_SIZE"][n] * 1024 + vals["WRITE_SIZE"][n] * 1024) for n in names}
    note = ("measured by this invocation: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child runs (" + what + "); "
            "read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE: "
            + "; ".join("%s %d + %d bytes" % (n, int(2 * vals["FETCH_SIZE"][n] * 1024), int(vals["WRITE_SIZE"][n] * 1024)) for n in names))
    if not many:
        note = note.replace(names[0] + " ", "")
        return byts[names[0]], note
    return byts, note


def single_process(a, t_start):
    """``--gpus N --single-process``: the N-GPU step driven from ONE host process through the library's own multi-GPU driver
    (``twx_multi_*``: one context + host thread per device, one ncclAllGather of the records, RCCL bound by the library) — the
    route a MATLAB / Octave / C host has.  Same workload, same timing rules and the same ``collective`` object as the
    one-process-per-GPU form; a box with fewer GPUs than N repeats its devices (records then concatenated on the host)."""
    import numpy as np
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    from amaranth_twstft_amd import _lib as L, prn
    from amaranth_twstft_amd.correlator import band_godual
    from amaranth_twstft_amd.multi import MultiCorrelator
    lib = L.load()
    world, nwin = a.gpus, a.windows
    devices = [i % ndev for i in range(world)]
    chips = prn.lfsr_chips(BITLEN, TAPS, NCHIPS)
    m = MultiCorrelator(chips, devices, fs=FS, Nint=1, max_batch=a.batch)
    recs, delays = [], []
    for r in range(world):
        dev = torch.device("cuda", devices[r])
        torch.cuda.set_device(dev)
        chips_dev = torch.from_numpy(chips).to(dev)
        iq = torch.empty((nwin, N, 2), dtype=torch.int16, device=dev)
        dl = []
        for p in range(nwin):
            sp, delay = window_params(p, r)
            dl.append(delay)
            params = np.array([sp.delay_q8, sp.fstep, sp.phi0, sp.amp, sp.noise_gain, sp.seed, sp.stream, 0], dtype=np.int64)
            L.check(lib.twx_synth_capture_dev(iq[p].data_ptr(), N, 0, chips_dev.data_ptr(), NCHIPS, 2, 1, params.ctypes.data_as(C.c_void_p), None))
        torch.cuda.synchronize(dev)
        recs.append(iq); delays.append(dl)
    ptrs = [t.data_ptr() for t in recs]
    band = band_godual(FS, N)
    df_true = np.full(nwin, 1780.75)

    def step(workload, fetch=False):
        return m.process_dev(ptrs, nwin, band=band if workload == "processing" else None, df=None if workload == "processing" else df_true, fetch=fetch)

    def timed(workload):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(workload)                               # returns after the gather: every device's stream is idle
        return time.perf_counter() - t0

    t_ready = time.perf_counter()
    for _ in range(a.warmup):
        step(a.workload)
    dt = timed(a.workload)
    other = "xcorr" if a.workload == "processing" else "processing"
    step(other)
    dt_other = timed(other)
    # --- BASELINE.json configs[3] as written: ONE nwin-window recording sharded over the contexts (twx_multi_process_recording_dev)
    strong = None
    if world > 1:
        blocks = [m.block(nwin, r) for r in range(world)]
        cap = max(c for _, c in blocks)
        sblk = []
        for r, (s0, c0) in enumerate(blocks):
            devr = torch.device("cuda", devices[r])
            torch.cuda.set_device(devr)
            cd = torch.from_numpy(chips).to(devr)
            t = torch.empty((max(c0, 1), N, 2), dtype=torch.int16, device=devr)
            for i in range(c0):
                sp, _ = window_params(s0 + i, 0)
                params = np.array([sp.delay_q8, sp.fstep, sp.phi0, sp.amp, sp.noise_gain, sp.seed, sp.stream, 0], dtype=np.int64)
                L.check(lib.twx_synth_capture_dev(t[i].data_ptr(), N, 0, cd.data_ptr(), NCHIPS, 2, 1, params.ctypes.data_as(C.c_void_p), None))
            torch.cuda.synchronize(devr)
            sblk.append(t)
        sptr = [t.data_ptr() for t in sblk]
        sstep = lambda fetch=False: m.process_recording_dev(sptr, nwin, band=band if a.workload == "processing" else None,
                                                            df=None if a.workload == "processing" else df_true, fetch=fetch)
        for _ in range(max(1, a.warmup)):
            sstep()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            sstep()
        sdt = time.perf_counter() - t0
        gms = [m.exchange_only(cap) for _ in range(max(5, a.steps))]
        sgot = sstep(True)
        sarr = (L.twx_result * nwin).from_buffer_copy(sgot.tobytes())
        sinfo = m.info
        s_rank = [all(int(sarr[s0 + i].indice0) == 3 * window_params(s0 + i, 0)[1] for i in range(c0)) for s0, c0 in blocks]
        # every context's device copy of the (padded) gather holds every rank's block
        s_copies = []
        for r in range(world):
            g = m.fetch_gathered(r, world * cap).reshape(world, cap, -1)
            s_copies.append(all(g[q, :blocks[q][1]].tobytes() == sgot[blocks[q][0]:blocks[q][0] + blocks[q][1]].tobytes() for q in range(world)))
        svalue = nwin * N * a.steps / sdt / 1e6
        strong = {"workload": f"BASELINE.json configs[3] as written: ONE {nwin}-window recording sharded over {world} contexts (contiguous blocks of "
                              f"{min(c for _, c in blocks)}..{cap} windows, HBM-resident), "
                              + ("processing(d,k) full chain" if a.workload == "processing" else "xcorr, df supplied")
                              + ", one exchange of the 240-byte records per step (twx_multi_process_recording_dev)",
                  "scaling": "strong", "value": round(svalue, 2), "unit": "Msamples/s", "ms_per_step": round(sdt / a.steps * 1e3, 3), "steps": a.steps,
                  "windows_total": nwin, "windows_per_rank": [c for _, c in blocks],
                  "gather_ms": round(float(np.median(gms)), 3), "gather_ms_note": "the exchange alone (twx_multi_exchange_only: the compute removed), median of %d" % len(gms),
                  "ratio_to_weak_headline": round(svalue / (world * nwin * N * a.steps / dt / 1e6), 4),
                  "collective": {"backend": "rccl" if sinfo.rccl else "host", "bytes_per_rank": int(sinfo.bytes_per_rank), "records": int(sinfo.records_gathered)},
                  "ranks_with_exact_lags": int(sum(s_rank)), "all_ranks_agree": bool(all(s_rank) and all(s_copies))}
        del sblk
    got = step(a.workload, fetch=True)
    info = m.info
    arr = (L.twx_result * (world * nwin)).from_buffer_copy(got.tobytes())
    per_rank = [all(int(arr[r * nwin + p].indice0) == 3 * delays[r][p] for p in range(nwin)) for r in range(world)]
    copies = [m.fetch_gathered(r, world * nwin).tobytes() == got.tobytes() for r in range(world)]
    value = world * nwin * N * a.steps / dt / 1e6
    out = {"metric": "Msamples/s correlated (1 s integrations, 2.5 Mchip PRN)", "value": round(value, 2), "unit": "Msamples/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("processing(d,k) full chain" if a.workload == "processing" else "xcorr, df supplied")
                      + ": the window of BASELINE.json configs[1] (1 s, 5 Msps int16 IQ, 2.5 Mchip LFSR(22,3) code, fp32), Nint=1, "
                      + f"one {nwin}-window recording per GPU per step, HBM-resident; ONE host process driving {world} contexts "
                      + f"on devices {devices} through twx_multi_* (one host thread per context)",
                      "windows_per_gpu_per_step": nwin, "samples_per_window": N, "sharding": f"windows/{world}", "launch": "single process",
                      "devices": devices, "timed_region_s": round(dt, 3)},
           "integer_lag_exact": bool(all(per_rank)),
           "startup_s": {"to_first_step": round(t_ready - t_start, 2)},
           "other_workload": {"workload": "xcorr, df supplied (code-phase only)" if other == "xcorr" else "processing(d,k) full chain",
                              "value": round(world * nwin * N * a.steps / dt_other / 1e6, 2), "unit": "Msamples/s",
                              "ms_per_step": round(dt_other / a.steps * 1e3, 3)},
           "collective": {"op": "ncclAllGather (RCCL %d, ncclCommInitAll, one group call from the host process)" % info.rccl_version if info.rccl
                                else ("host-side concatenation (RCCL given up, see backend)" if info.rccl_fallback else
                                      "host-side concatenation (the device list repeats a device: RCCL has one rank per device)"),
                          "backend": "rccl" if info.rccl else ("host (fallback: %s)" % info.rccl_error.decode(errors="replace") if info.rccl_fallback else "host"),
                          "rccl_fallback": int(info.rccl_fallback), "world": world, "records": int(info.records_gathered),
                          "threads_pinned": int(info.threads_pinned), "numa_nodes": [int(info.numa_node[i]) for i in range(world)],
                          "bytes_per_rank": int(info.bytes_per_rank), "gather_ms_last": round(info.gather_ms, 3),
                          "ranks_with_exact_lags": int(sum(per_rank)), "gathered_lag_exact": bool(all(per_rank)),
                          "own_block_identical": bool(all(copies)), "all_ranks_agree": bool(all(copies) and all(per_rank)),
                          "note": "every context's device copy of the gathered buffer was fetched and compared"}}
    if strong is not None:
        out["strong_workload"] = strong
    print(json.dumps(out))
    m.close()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: two stations x {local, remote} = four correlations on a 70-Msps wideband capture, fp32 and fp64
# (acquisition/go_1s.m:88,120,147,171 consumes exactly these four series)
# ------------------------------------------------------------------------------------------------------------------
ALGO_BYTES_F64 = {"k_sums": lambda S: 4 * S, "k_col_fwd_square": lambda S: 20 * S, "k_row_band": lambda S: 16 * S, "k_df_tables": lambda S: 0,
                  "k_col_fwd_mix": lambda S: 20 * S, "k_row_mid": lambda S: 64 * S + 16 * N, "k_col_inv": lambda S: 48 * S, "k_peak": lambda S: 0}


def wideband_legs(local_rank: int, seconds: int = 4, steps: int = 5):
    """Per step: for each of the two stations a `seconds`-long 70-Msps int16 capture (own code in the loop-back + the other station's
    code 50 kHz off, chips held 28 samples) -> twx_fir_decimate_dev (Hamming low-pass of frontend.lowpass_taps: 421 taps, decimate by 14) -> 5-Msps int16 ->
    FOUR correlations in flight together (four contexts, each with its own streams): OPlo, OPre, LTFBlo, LTFBre, every one the full
    processing(d,k) over `seconds` 1-s windows with its own band.  Returns (wideband_workload, f64_workload): the fp32 chain with the
    FIR's own roofline, and the same four correlations in fp64 with the roofline of their dominant kernel and the fp32-vs-fp64
    peak-magnitude comparison configs[4] asks for."""
    import numpy as np
    import torch
    from amaranth_twstft_amd import _lib as L, frontend, prn, synth
    from amaranth_twstft_amd.correlator import Correlator, band_godual
    from amaranth_twstft_amd.wideband import WidebandSession
    lib = L.load()
    dev = torch.device("cuda", local_rank)
    fs_in, dec, sps_in = 70e6, 14, 28
    taps = frontend.lowpass_taps(fs_in, 2.1e6, 0.4e6)
    ntaps, W = int(taps.size), int(seconds)
    n_out = W * N
    n_in = (n_out - 1) * dec + ntaps
    half = (ntaps - 1) // 2
    codes = {"OP": prn.lfsr_chips(BITLEN, 57, NCHIPS), "LTFB": prn.lfsr_chips(BITLEN, 3, NCHIPS)}
    cdev = {k: torch.from_numpy(v).to(dev) for k, v in codes.items()}
    # (station, other, local delay [70-Msps samples], remote delay, remote carrier offset, OP flag of the remote band)
    stations = [("OP", "LTFB", 18_364_717, 50_772_133, +50_000.0, 0), ("LTFB", "OP", 41_000_003, 9_123_457, -50_000.0, 1)]

    def synth_into(out, chips_dev, p):
        params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(out.data_ptr(), n_in, 0, chips_dev.data_ptr(), NCHIPS, sps_in, 1, params.ctypes.data_as(C.c_void_p), None))

    wide, expect = {}, {}
    for i, (st, other, d_loc, d_rem, f_rem, op_flag) in enumerate(stations):
        a_ = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        b_ = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        synth_into(a_, cdev[st], synth.SynthParams(delay_q8=d_loc * 256, fstep=synth.fstep_for_df(3.25, fs_in), phi0=99, amp=2500,
                                                   noise_gain=synth.noise_gain_for_sigma(2500.0), seed=401, stream=2 * i))
        synth_into(b_, cdev[other], synth.SynthParams(delay_q8=d_rem * 256, fstep=synth.fstep_for_df(f_rem, fs_in), phi0=7, amp=1200,
                                                      noise_gain=0, seed=402, stream=2 * i))
        torch.cuda.synchronize(dev)
        step_ = 1 << 26
        for lo in range(0, n_in, step_):                      # a + b with saturation, in pieces (no 4-byte copy of the whole capture)
            sl = slice(lo, min(n_in, lo + step_))
            a_[sl] = (a_[sl].to(torch.int32) + b_[sl].to(torch.int32)).clamp_(-32768, 32767).to(torch.int16)
        del b_
        wide[st] = a_
        expect[st + "lo"], expect[st + "re"] = (d_loc - half) / dec, (d_rem - half) / dec
    torch.cuda.synchronize(dev)                                # torch's stream wrote the captures; the library's streams do not wait for it
    # name -> (capture's station, code, band)
    plan = {"OPlo": ("OP", "OP", band_godual(FS, N)), "OPre": ("OP", "LTFB", band_godual(FS, N, remote=1, OP=0)),
            "LTFBlo": ("LTFB", "LTFB", band_godual(FS, N)), "LTFBre": ("LTFB", "OP", band_godual(FS, N, remote=1, OP=1))}
    bands = {k: L.twx_band(*v[2]) for k, v in plan.items()}
    res = {(k, pr): torch.zeros((W, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev) for k in plan for pr in ("f64",)}
    wptr = {st: t.data_ptr() for st, t in wide.items()}

    def records_of(step_records):
        return {k: [(int(r.indice0), math.hypot(r.xval[0], r.xval[1]), float(r.df)) for r in v] for k, v in step_records.items()}

    # fp32: the product's session (amaranth_twstft_amd/wideband.py): front end + four correlations, double-buffered by step
    sess = WidebandSession(codes, taps, dec, fs=FS, windows=W, device=local_rank, precision="f32", plan=plan)
    c32 = sess.ctx

    def fir_all():
        for st in sess.stations:
            c32[sess.front[st]].fir_decimate_dev(wptr[st], n_in, taps, dec, out_i16_dev=sess.nar[st][0].data_ptr())
        for st in sess.stations:
            c32[sess.front[st]].synchronize()

    def corr_all32():
        for k, (st, _, _) in plan.items():                     # four launches back to back: nothing waits in between
            L.check(lib.twx_process_windows_dev(c32[k]._h, sess.nar[st][0].data_ptr(), W, 1, 0, C.byref(bands[k]), None, sess.res[k][0].data_ptr()), c32[k]._h)
        for k in plan:
            c32[k].synchronize()

    sess.submit(wptr); sess.submit(wptr); sess.synchronize()    # warm-up (both buffer parities)
    t0 = time.perf_counter()
    for _ in range(steps):
        fir_all()
    t_fir = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        corr_all32()
    t_corr = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):                                      # one step at a time: submit, wait
        sess.submit(wptr); sess.synchronize()
    t_step_serial = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    last = -1
    for _ in range(steps):                                      # the session as it is meant to run: step i+1 enqueued behind step i,
        last = sess.submit(wptr)                                # its front end sharing the GPU with step i's correlations
        if last >= 1:
            sess.fetch(last - 1)                                # (and every step's records fetched, one step behind)
    r32 = records_of(sess.fetch(last))
    sess.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    # the FIR kernel alone: HIP events on the stream it is launched on (the context's, not torch's)
    es = sess.stream["OPlo"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c32["OPlo"].fir_decimate_dev(wptr["OP"], n_in, taps, dec, out_i16_dev=sess.nar["OP"][0].data_ptr())
    e0.record(es)
    for _ in range(5):
        c32["OPlo"].fir_decimate_dev(wptr["OP"], n_in, taps, dec, out_i16_dev=sess.nar["OP"][0].data_ptr())
    e1.record(es)
    e1.synchronize()
    fir_ms = e0.elapsed_time(e1) / 5
    fir_flops = n_out * ntaps * 4                                # complex int16 sample x real tap: 2 FMAs
    fir_bytes = n_in * 4 + n_out * 4
    sess.synchronize()
    # the opt-in matrix-core form of the front end, ALONE on the GPU (nothing else in flight: profiles/r05_fir_mfma.txt), against the vector form's output
    fir_mc = None
    try:
        vec_out = sess.nar["OP"][0].clone()
        L.check(lib.twx_set_option(c32["OPlo"]._h, L.TWX_OPT_FIR_MFMA, 1), c32["OPlo"]._h)     # (the session set it to 0 on its contexts)
        c32["OPlo"].fir_decimate_dev(wptr["OP"], n_in, taps, dec, out_i16_dev=sess.nar["OP"][0].data_ptr())
        e0.record(es)
        for _ in range(5):
            c32["OPlo"].fir_decimate_dev(wptr["OP"], n_in, taps, dec, out_i16_dev=sess.nar["OP"][0].data_ptr())
        e1.record(es)
        e1.synchronize()
        mc_ms = e0.elapsed_time(e1) / 5
        same = int((sess.nar["OP"][0].to(torch.int32) - vec_out.to(torch.int32)).abs().max().item())
        fir_mc = {"kernel": "k_fir_mfma (fp16 matrix cores, samples and taps split exactly; TWX_OPT_FIR_MFMA = 1)", "avg_ms": round(mc_ms, 4),
                  "input_Msamples_per_s": round(n_in / mc_ms / 1e3, 1), "GB/s": round(fir_bytes / (mc_ms * 1e-3) / 1e9, 1),
                  "frac_hbm": round(fir_bytes / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "fp32_equivalent_TFLOP/s": round(fir_flops / (mc_ms * 1e-3) / 1e12, 2), "frac_of_fp32_vector_peak": round(fir_flops / (mc_ms * 1e-3) / 1e12 / 157.3, 4),
                  "max_abs_difference_from_the_vector_form_int16": same,
                  "note": "opt-in (TWX_OPT_FIR_MFMA); packed-fp32 results of waves resident beside this kernel go wrong (profiles/r05_fir_mfma.txt), so the "
                          "library orders every such launch behind all other work it has enqueued on the device and all later work behind it "
                          "(csrc/twx_internal.h): it always runs alone.  No timed step of this line uses it"}
    finally:
        L.check(lib.twx_set_option(c32["OPlo"]._h, L.TWX_OPT_FIR_MFMA, 0), c32["OPlo"]._h)
        c32["OPlo"].fir_decimate_dev(wptr["OP"], n_in, taps, dec, out_i16_dev=sess.nar["OP"][0].data_ptr())      # the vector form's output for what follows
        sess.synchronize()
    nar = {st: sess.nar[st][0].clone() for st in sess.stations}   # the decimated captures for the fp64 leg
    torch.cuda.synchronize(dev)
    sess.close()

    def make(precision, **kw):
        return {k: Correlator(codes[v[1]], fs=FS, Nint=1, device=local_rank, precision=precision, **kw) for k, v in plan.items()}

    def corr_all(ctx, pr):
        for k, (st, _, _) in plan.items():                     # four launches back to back: nothing waits in between
            L.check(lib.twx_process_windows_dev(ctx[k]._h, nar[st].data_ptr(), W, 1, 0, C.byref(bands[k]), None, res[(k, pr)].data_ptr()), ctx[k]._h)
        for k in plan:
            ctx[k].synchronize()

    def records(pr):
        out = {}
        for k in plan:
            arr = (L.twx_result * W).from_buffer_copy(res[(k, pr)].cpu().numpy().tobytes())
            out[k] = [(int(arr[w].indice0), math.hypot(arr[w].xval[0], arr[w].xval[1]), float(arr[w].df)) for w in range(W)]
        return out

    c64 = make("f64")
    corr_all(c64, "f64")
    t0 = time.perf_counter()
    for _ in range(steps):
        corr_all(c64, "f64")
    t_corr64 = (time.perf_counter() - t0) / steps
    r64 = records("f64")
    for c in c64.values():
        c.close()
    pc = Correlator(codes["OP"], fs=FS, Nint=1, device=local_rank, precision="f64", profile=True)
    run1 = lambda: L.check(lib.twx_process_windows_dev(pc._h, nar["OP"].data_ptr(), W, 1, 0, C.byref(bands["OPlo"]), None, res[("OPlo", "f64")].data_ptr()), pc._h)
    run1(); pc.profile(reset=True)
    for _ in range(3):
        run1()
    prof = pc.profile()
    pc.close()
    kern = {k: (v["ms_total"] / v["launches"], v["units"] / v["launches"]) for k, v in prof.items() if v["launches"]}
    dom = max(kern, key=lambda k: prof[k]["ms_total"])
    dbytes = ALGO_BYTES_F64[dom](kern[dom][1])
    dach = dbytes / (kern[dom][0] * 1e-3) / 1e9

    lag_ok = all(abs(((r32[k][w][0] / 3.0 - expect[k] + N / 2) % N) - N / 2) < 1.0 for k in plan for w in range(W))
    same = all(r32[k][w][0] == r64[k][w][0] for k in plan for w in range(W))
    rel = max(abs(r32[k][w][1] - r64[k][w][1]) / r64[k][w][1] for k in plan for w in range(W))
    corr_samples = 4 * W * N
    wl = {"workload": f"BASELINE.json configs[4]: two stations x {{local, remote}} = 4 correlations (OPlo, OPre, LTFBlo, LTFBre: two LFSR(22) codes, taps 57 / 3, "
                      f"remote signal +-50 kHz off), {W} s of 70-Msps int16 IQ per station, HBM-resident: FIR {ntaps} taps decimate by 14 -> 5 Msps -> full "
                      "processing(d,k) per 1-s window, the four correlations in flight together (four contexts), fp32; steps double-buffered "
                      "(amaranth_twstft_amd/wideband.py: step i+1 is enqueued behind step i, every step's records fetched one step behind)",
          "input_Msamples_per_s": round(2 * n_in / t_step / 1e6, 1), "correlated_Msamples_per_s": round(corr_samples / t_step / 1e6, 1),
          "ms_per_step": round(t_step * 1e3, 3), "ms_per_step_one_at_a_time": round(t_step_serial * 1e3, 3),
          "input_Msamples_per_s_one_at_a_time": round(2 * n_in / t_step_serial / 1e6, 1),
          "fir_ms_per_step": round(t_fir * 1e3, 3), "correlations_ms_per_step": round(t_corr * 1e3, 3),
          "correlations_alone_Msamples_per_s": round(corr_samples / t_corr / 1e6, 1),
          "chain_GBs_algorithmic": round(92 * corr_samples / t_corr / 1e9, 1), "chain_frac_hbm": round(92 * corr_samples / t_corr / 1e9 / HBM_PEAK_GBS, 4),
          "expected_lags_within_one_sample": bool(lag_ok), "steps": steps,
          "fir": {"kernel": "k_fir_poly8", "avg_ms": round(fir_ms, 4), "input_Msamples_per_s": round(n_in / fir_ms / 1e3, 1),
                  "roofline": {"bound": "fp32 vector", "achieved": round(fir_flops / (fir_ms * 1e-3) / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                               "frac": round(fir_flops / (fir_ms * 1e-3) / 1e12 / 157.3, 4), "flops_per_launch": int(fir_flops)},
                  "GB/s": round(fir_bytes / (fir_ms * 1e-3) / 1e9, 1), "frac_hbm": round(fir_bytes / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "timing": "HIP events on the context's stream around 5 launches", "matrix_core_form": fir_mc}}
    f64 = {"workload": f"the same four correlations in fp64 (twx_config.precision = f64) on the decimated captures, {W} windows each, in flight together",
           "dtype": "f64", "correlated_Msamples_per_s": round(corr_samples / t_corr64 / 1e6, 1), "ms_per_step": round(t_corr64 * 1e3, 3),
           "chain_GBs_algorithmic": round(180 * corr_samples / t_corr64 / 1e9, 1), "chain_frac_hbm": round(180 * corr_samples / t_corr64 / 1e9 / HBM_PEAK_GBS, 4),
           "roofline": {"bound": "hbm", "kernel": dom + " (fp64)", "achieved": round(dach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dach / HBM_PEAK_GBS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": int(dbytes), "avg_ms": round(kern[dom][0], 4), "launches_timed": int(prof[dom]["launches"]),
                        "note": "one correlation alone in a one-slot profiling context; complex double = 16 B"},
           "kernels": {k: {"avg_ms": round(v[0], 4), "GB/s": round(ALGO_BYTES_F64[k](v[1]) / (v[0] * 1e-3) / 1e9, 1)} for k, v in kern.items() if k in ALGO_BYTES_F64},
           "integer_lags_equal_fp32": bool(same), "fp32_vs_fp64_peak_rel": float("%.3g" % rel), "tolerance": 1e-6,
           "within_tolerance": bool(rel <= 1e-6 and same)}
    return wl, f64


def strong_leg(a, ex, cor, lib, L, dev, rank, world, chips_dev, band, df_true, weak_value):
    """BASELINE.json configs[3] as written (godual_ranging.m:75-102: ONE recording, consecutive windows): `--windows`
    windows IN TOTAL, rank r owning the contiguous block dist.shard_windows names, per step its block through the full chain, one
    all-gather of the fixed-size records (blocks padded to the longest), the same barrier + max-over-ranks timing as the headline.
    Reported beside the weak headline, never instead of it: this is the leg that can show a scaling defect (per-step synchronise +
    gather + barrier against 1/N of the compute, the launch tail of a short block).  Also times the exchange alone (compute removed)."""
    import numpy as np
    import torch
    from amaranth_twstft_amd import dist
    total = a.windows
    lo, hi = dist.shard_windows(total, rank, world)
    nloc, cap = hi - lo, dist.max_shard(total, world)
    RB = C.sizeof(L.twx_result)
    iq_s = torch.empty((max(nloc, 1), N, 2), dtype=torch.int16, device=dev)
    for i, p in enumerate(range(lo, hi)):                     # ONE recording description: window p of rank 0's parameters, whoever holds it
        sp, _ = window_params(p, 0)
        params = np.array([sp.delay_q8, sp.fstep, sp.phi0, sp.amp, sp.noise_gain, sp.seed, sp.stream, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(iq_s[i].data_ptr(), N, 0, chips_dev.data_ptr(), NCHIPS, 2, 1, params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    res_s = torch.zeros((cap, RB), dtype=torch.uint8, device=dev)
    gat_s = torch.zeros((world * cap, RB), dtype=torch.uint8, device=dev)
    dfp = np.ascontiguousarray(df_true[:1].repeat(max(nloc, 1)))

    def compute():
        if nloc:
            if a.workload == "processing":
                L.check(lib.twx_process_windows_dev(cor._h, iq_s.data_ptr(), nloc, 1, 0, C.byref(band), None, res_s.data_ptr()), cor._h)
            else:
                L.check(lib.twx_process_windows_dev(cor._h, iq_s.data_ptr(), nloc, 1, 0, None, dfp.ctypes.data_as(C.c_void_p), res_s.data_ptr()), cor._h)

    def exchange():
        L.check(lib.twx_synchronize(cor._h), cor._h)          # records complete before the exchange reads them
        ex.all_gather_records(gat_s, res_s)

    def barrier():
        ex.barrier()
        L.check(lib.twx_synchronize(cor._h), cor._h)
        torch.cuda.synchronize()

    def timed(fn):
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        barrier()
        return ex.max_float(time.perf_counter() - t0)

    def step():
        compute()
        exchange()

    for _ in range(max(1, a.warmup)):
        step()
    dt = timed(step)
    dt_compute = timed(lambda: (compute(), L.check(lib.twx_synchronize(cor._h), cor._h)))      # the same blocks without the exchange
    # the exchange alone: every rank enters it together (barrier), the slowest rank's median
    gms = []
    for _ in range(max(5, a.steps)):
        ex.barrier()
        t0 = time.perf_counter()
        exchange()
        gms.append((time.perf_counter() - t0) * 1e3)
    gather_ms = ex.max_float(float(np.median(gms)))
    step()
    barrier()
    # every rank checks ITS copy of the gathered recording against the generator: all `total` lags, in window order
    gh = gat_s.cpu().numpy().reshape(world, cap, RB)
    per_rank = []
    for r in range(world):
        s0, e0 = dist.shard_windows(total, r, world)
        arr = (L.twx_result * max(e0 - s0, 1)).from_buffer_copy(np.ascontiguousarray(gh[r, :max(e0 - s0, 1)]).tobytes())
        per_rank.append(bool(all(int(arr[i].indice0) == 3 * window_params(s0 + i, 0)[1] for i in range(e0 - s0))))
    own = bool(gh[rank, :nloc].tobytes() == res_s[:nloc].cpu().numpy().tobytes())
    agree = ex.all_true(bool(all(per_rank) and own))
    value = total * N * a.steps / dt / 1e6
    blocks = [dist.shard_windows(total, r, world) for r in range(world)]
    return {"workload": f"BASELINE.json configs[3] as written: ONE {total}-window recording sharded over {world} ranks (contiguous blocks of "
                        f"{min(e - s for s, e in blocks)}..{max(e - s for s, e in blocks)} windows, HBM-resident), "
                        + ("processing(d,k) full chain" if a.workload == "processing" else "xcorr, df supplied")
                        + ", one all-gather of the 240-byte records per step",
            "scaling": "strong", "value": round(value, 2), "unit": "Msamples/s", "ms_per_step": round(dt / a.steps * 1e3, 3), "steps": a.steps,
            "windows_total": total, "windows_per_rank": [e - s for s, e in blocks],
            "compute_only_ms_per_step": round(dt_compute / a.steps * 1e3, 3),
            "gather_ms": round(gather_ms, 3), "gather_ms_note": "the exchange alone (synchronise + all-gather of the padded blocks, every rank entering "
                                                               "together), median of %d, the slowest rank's" % len(gms),
            "exchange_share_of_step": round(max(0.0, 1.0 - dt_compute / dt), 4),
            "ratio_to_weak_headline": round(value / weak_value, 4),
            "ratio_note": "strong value / weak value of the same run: 1.0 = a rank processes its 1/N block at the rate it processes a whole "
                          "recording (per-step overheads hidden); the N = 1 rate is the driver's own N = 1 line",
            "collective": {"backend": ex.describe(), "bytes_per_rank": cap * RB, "records": world * cap},
            "ranks_with_exact_lags": int(sum(per_rank)), "own_block_identical": own, "all_ranks_agree": bool(agree)}



def caf_only():
    """Child of the CAF's `rocprofv3 --pmc` passes: one synthetic window, 128 Doppler bins = two full 64-bin launches of k_rowd_caf
    and of the last pass; nothing else of the bench runs."""
    import numpy as np
    import torch
    from amaranth_twstft_amd import _lib as L, prn
    from amaranth_twstft_amd.correlator import Correlator
    lib = L.load()
    dev = torch.device("cuda", 0)
    chips = prn.lfsr_chips(BITLEN, TAPS, NCHIPS)
    sp, _ = window_params(0, 0)
    iq = torch.empty((N, 2), dtype=torch.int16, device=dev)
    params = np.array([sp.delay_q8, sp.fstep, sp.phi0, sp.amp, sp.noise_gain, sp.seed, sp.stream, 0], dtype=np.int64)
    L.check(lib.twx_synth_capture_dev(iq.data_ptr(), N, 0, torch.from_numpy(chips).to(dev).data_ptr(), NCHIPS, 2, 1, params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    with Correlator(chips, fs=FS, Nint=0, device=0) as cc:
        cc.caf_bins_dev(iq.data_ptr(), 1717, 1717 + 127)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=600, help="1-s windows per GPU per step (resident in HBM; 600 = a 10-minute recording)")
    ap.add_argument("--batch", type=int, default=0, help="channel-windows per launch (0 = library default)")
    ap.add_argument("--workload", choices=["processing", "xcorr"], default="processing",
                    help="processing = full processing(d,k); xcorr = df supplied (code-phase-only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=6, help="windows of the workload timed on one host core (oracle)")
    ap.add_argument("--cpu-max-workers", type=int, default=64)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc child passes (roofline.traffic then comes "
                    "from the committed profiles/pmc_traffic.json)")
    ap.add_argument("--no-caf", action="store_true", help="skip the BASELINE.json configs[2] leg (delay x Doppler CAF of one window)")
    ap.add_argument("--caf-only", action="store_true", help=argparse.SUPPRESS)      # the child of the CAF's --pmc passes: two full launches of the surface
    ap.add_argument("--no-wideband", action="store_true", help="skip the BASELINE.json configs[4] legs (70-Msps front end + four correlations, fp32 and fp64)")
    ap.add_argument("--wideband-only", action="store_true", help="run only the configs[4] legs and print their two objects (profiling runs)")
    ap.add_argument("--wideband-seconds", type=int, default=4, help="seconds of 70-Msps capture per station in the configs[4] legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                    "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-process", action="store_true", help="drive the --gpus N devices from THIS process through the library's "
                    "twx_multi driver (host threads + RCCL inside the library) instead of one rank per GPU under torch.distributed.run")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the collectives even with one "
                    "rank (exercises the RCCL calls of the N > 1 path on a one-GPU box)")
    a = ap.parse_args()
    t_start = time.perf_counter()

    if a.cpu_baseline_child:
        cpu_baseline_child(a.cpu_windows, a.workload, a.cpu_max_workers)
        return

    if a.single_process:
        single_process(a, t_start)
        return
    if a.caf_only:
        caf_only()
        return
    if a.wideband_only:
        wl, f64 = wideband_legs(0, a.wideband_seconds)
        print(json.dumps({"wideband_workload": wl, "f64_workload": f64}))
        return

    from amaranth_twstft_amd import collective, launch
    if a.gpus > 1 and not launch.is_rank():
        # not started by torchrun: start the N ranks as a child job (nothing here has touched the GPU yet); a job that ends with
        # a non-zero status is started once more, fresh, with the exchange on gloo — the line then says so
        sys.exit(launch.spawn_with_fallback(a.gpus, os.path.abspath(__file__), sys.argv[1:], backend=a.backend))
    rank, local_rank, world = launch.rank_world()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE {world}: start with --nproc-per-node {a.gpus}, or without torchrun")
    # control plane (gloo) and the RCCL probe job: before this process touches the GPU (amaranth_twstft_amd/collective.py)
    use_dist = world > 1 or a.force_dist
    ex = collective.RecordExchange(rank, world, want=a.backend, reason=os.environ.get("TWX_COLLECTIVE_FALLBACK_REASON") or None)
    if use_dist:
        ex.prepare(force=a.force_dist)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = run_cpu_baseline(a.cpu_windows, a.workload, a.cpu_max_workers)      # before the first GPU call
    pmc_live = pmc_caf = None
    if rank == 0 and world == 1 and not a.no_roofline and not a.no_pmc and a.workload == "processing":
        t_p = time.perf_counter()
        pmc_live = measure_pmc_traffic("k_row_mid") + (round(time.perf_counter() - t_p, 1),)       # child processes, before the first GPU call
        if not a.no_caf:
            t_p = time.perf_counter()
            pmc_caf = measure_pmc_traffic(("k_rowd_caf", "k_col_inv"), ("--caf-only",), "128 Doppler bins = two full launches of each kernel") \
                + (round(time.perf_counter() - t_p, 1),)

    import numpy as np
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    sharing = world > ndev                  # fewer GPUs than ranks (tests): ranks share devices, the exchange is then gloo by necessity
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from amaranth_twstft_amd import _lib as L, prn
    from amaranth_twstft_amd.correlator import Correlator, band_godual
    lib = L.load()
    pin = collective.pin_to_device(lib, local_rank)         # this rank's threads next to its GPU (NUMA node of the device)

    chips = prn.lfsr_chips(BITLEN, TAPS, NCHIPS)
    cor = Correlator(chips, fs=FS, Nint=1, device=local_rank, max_batch=a.batch)
    nwin = a.windows
    # --- synthetic recording, generated on the device, resident in HBM
    chips_dev = torch.from_numpy(chips).to(dev)
    iq = torch.empty((nwin, N, 2), dtype=torch.int16, device=dev)
    delays = []
    for p in range(nwin):
        sp, delay = window_params(p, rank)
        delays.append(delay)
        params = np.array([sp.delay_q8, sp.fstep, sp.phi0, sp.amp, sp.noise_gain, sp.seed, sp.stream, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(iq[p].data_ptr(), N, 0, chips_dev.data_ptr(), NCHIPS, 2, 1,
                                          params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    res = torch.zeros((nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world * nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev) if use_dist else None
    band = L.twx_band(*band_godual(FS, N))
    df_true = np.array([1780.75] * nwin, dtype=np.float64)
    if use_dist:
        # data plane: RCCL sub-group + a rehearsal of the real gather, each under a deadline and agreed on by all ranks; any
        # failure leaves the exchange on gloo, flagged in the line
        ex.bring_up(dev, rehearsal=(gathered, res))

    def step(c, workload=None):
        if (workload or a.workload) == "processing":
            L.check(lib.twx_process_windows_dev(c._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), c._h)
        else:
            L.check(lib.twx_process_windows_dev(c._h, iq.data_ptr(), nwin, 1, 0, None, df_true.ctypes.data_as(C.c_void_p),
                                                res.data_ptr()), c._h)
        if use_dist:
            L.check(lib.twx_synchronize(c._h), c._h)        # results complete before the exchange reads them
            ex.all_gather_records(gathered, res)            # RCCL over xGMI, 240 B per window (returns when `gathered` is complete:
                                                            # the next step rewrites `res` from the library's own streams)

    def barrier():
        ex.barrier()
        L.check(lib.twx_synchronize(cor._h), cor._h)
        torch.cuda.synchronize()

    def timed(workload, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(cor, workload)
        barrier()
        return ex.max_float(time.perf_counter() - t0)      # the slowest rank's time

    t_ready = time.perf_counter()
    for _ in range(a.warmup):
        step(cor)
    dt = timed(a.workload, a.steps)

    # --- the other workload, timed the same way (reported beside `value`, never instead of it)
    other = "xcorr" if a.workload == "processing" else "processing"
    step(cor, other)
    dt_other = timed(other, a.steps)
    step(cor)                            # leave the results of the headline workload in `res` for the checks below
    barrier()

    # --- correctness of what was timed: integer lags must equal the generator's delays
    host = res.cpu().numpy().tobytes()
    arr = (L.twx_result * nwin).from_buffer_copy(host)
    lag_ok = all(int(arr[p].indice0) == 3 * delays[p] for p in range(nwin))

    # --- the collective moved the right data: every rank's block of the gathered buffer must hold THAT rank's lags
    # (window_params is a pure function of (window, rank), so rank 0 can recompute what every other rank was given)
    coll_info = None
    if use_dist:
        gh = gathered.cpu().numpy()
        garr = (L.twx_result * (world * nwin)).from_buffer_copy(gh.tobytes())
        per_rank = []
        for r in range(world):
            ok = all(int(garr[r * nwin + p].indice0) == 3 * window_params(p, r)[1] for p in range(nwin))
            per_rank.append(bool(ok))
        own = gh[rank * nwin:(rank + 1) * nwin].tobytes() == host
        coll_info = {"op": "all_gather_into_tensor", "backend": ex.describe(), "requested": a.backend, "world": world,
                      "records": world * nwin, "bytes_per_rank": nwin * C.sizeof(L.twx_result), "ranks_with_exact_lags": int(sum(per_rank)),
                      "gathered_lag_exact": bool(all(per_rank)), "own_block_identical": bool(own), "checked_on_rank": rank,
                      "control_plane": "gloo over 127.0.0.1 (barriers, max-over-ranks of the time, agreements)"}
        if ex.probe is not None:
            coll_info["rccl_probe"] = {k: ex.probe[k] for k in ("ok", "tried", "seconds") if k in ex.probe}
        if sharing:
            coll_info["ranks_sharing_devices"] = f"{world} ranks on {ndev} GPU(s)"
        # every rank checks its copy; rank 0 reports whether ALL copies were right
        coll_info["all_ranks_agree"] = ex.all_true(bool(all(per_rank) and own))
        coll_info["numa"] = ex.all_objects(pin)            # per rank: NUMA node of its GPU and the CPUs it was bound to

    # --- BASELINE.json configs[3] AS WRITTEN (N > 1 only; the weak headline above stays): ONE recording of `nwin` windows in total,
    # rank r processing its contiguous block (dist.shard_windows), one all-gather of the records per step
    strong = None
    if use_dist:
        strong = strong_leg(a, ex, cor, lib, L, dev, rank, world, chips_dev, band, df_true, weak_value=world * nwin * N * a.steps / dt / 1e6)

    samples = world * nwin * N * a.steps
    value = samples / dt / 1e6
    out = {"metric": "Msamples/s correlated (1 s integrations, 2.5 Mchip PRN)", "value": round(value, 2), "unit": "Msamples/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("processing(d,k) full chain" if a.workload == "processing" else "xcorr, df supplied")
                      + ": the window of BASELINE.json configs[1] (1 s, 5 Msps int16 IQ, 2.5 Mchip LFSR(22,3) code, fp32), Nint=1, "
                      + f"one {nwin}-window recording per GPU per step (configs[3]'s 10-minute recording at 600), HBM-resident"
                      + ("; configs[1]'s own code-phase-only xcorr is reported under other_workload" if a.workload == "processing" else ""),
                      "windows_per_gpu_per_step": nwin, "samples_per_window": N, "batch": int(cor.info.batch),
                      "n1": int(cor.info.n1), "n2": int(cor.info.n2), "sharding": f"windows/{world}",
                      "timed_region_s": round(dt, 3)},
           "integer_lag_exact": bool(lag_ok),
           "startup_s": {"to_first_step": round(t_ready - t_start, 2),
                         "note": "process start to the first warm-up step on this rank: imports, CPU baseline (N=1 only), context "
                                 "creation, synthesis of this rank's recording on its own GPU (ranks start up in parallel)"},
           "other_workload": {"workload": "xcorr, df supplied (code-phase only)" if other == "xcorr" else "processing(d,k) full chain",
                              "value": round(world * nwin * N * a.steps / dt_other / 1e6, 2), "unit": "Msamples/s",
                              "ms_per_step": round(dt_other / a.steps * 1e3, 3)}}

    # --- roofline of the dominant kernel: HIP events around every launch on the library's stream
    if rank == 0 and not a.no_roofline:
        pc = Correlator(chips, fs=FS, Nint=1, device=local_rank, max_batch=a.batch, profile=True)
        step_w = min(nwin, 192)

        def prof_pass():
            L.check(lib.twx_process_windows_dev(pc._h, iq.data_ptr(), step_w, 1, 0,
                                                C.byref(band) if a.workload == "processing" else None,
                                                None if a.workload == "processing" else df_true.ctypes.data_as(C.c_void_p),
                                                res.data_ptr()), pc._h)
        prof_pass()
        pc.profile(reset=True)          # discard the warm-up pass
        for _ in range(3):
            prof_pass()
        prof = pc.profile()
        kern = {k: dict(ms_avg=v["ms_total"] / v["launches"], launches=v["launches"],
                        samples_per_launch=v["units"] / v["launches"]) for k, v in prof.items()}
        dom = max(kern, key=lambda k: prof[k]["ms_total"])
        byts = ALGO_BYTES[dom](kern[dom]["samples_per_launch"])
        ach = byts / (kern[dom]["ms_avg"] * 1e-3) / 1e9
        out["roofline_note"] = ("per-kernel durations are the dispatches' own begin/end timestamps (HIP events attached to every launch) in a "
                                "context with ONE pipeline slot, i.e. kernels of different batches do not overlap; the timed region above runs "
                                "%d slots" % int(os.environ.get("TWX_STREAMS", "3")))
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                           "algorithmic_bytes_per_launch": int(byts), "avg_ms": round(kern[dom]["ms_avg"], 4),
                           "launches_timed": int(kern[dom]["launches"])}
        out["kernels"] = {k: {"avg_ms": round(v["ms_avg"], 4), "GB/s": round(ALGO_BYTES[k](v["samples_per_launch"]) / (v["ms_avg"] * 1e-3) / 1e9, 1)}
                          for k, v in kern.items()}
        tot = sum(ALGO_BYTES[k](kern[k]["samples_per_launch"]) for k in kern)
        spl = kern[dom]["samples_per_launch"]
        out["chain_GBs_algorithmic"] = round(tot / spl * value * 1e6 / 1e9 / world, 1)
        pc.close()
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if pmc_live and pmc_live[0] and dom == "k_row_mid":
            out["roofline"]["traffic"] = pmc_live[0]
            out["roofline"]["traffic_source"] = pmc_live[1] + " (%.0f s)" % pmc_live[2]
        elif os.path.exists(pmc):
            if pmc_live:
                out["roofline"]["traffic_live_error"] = pmc_live[1]
            try:
                t = json.load(open(pmc))
                if t.get("kernel") == dom:
                    out["roofline"]["traffic"] = t.get("bytes_per_launch")
                    out["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json (rocprofv3 --pmc passes of %s, not measured in this run)"
                                                         % t.get("source_commit", t.get("commit", "the commit named in that file")))
            except Exception:
                pass

    # --- BASELINE.json configs[2]: full delay x Doppler surface of ONE window, +-5 kHz at 1 Hz (10 001 bins of fs/N), with its own
    # roofline (reported beside the headline metric, never instead of it)
    if rank == 0 and world == 1 and not a.no_caf and not a.no_roofline:
        cc = Correlator(chips, fs=FS, Nint=0, device=local_rank)
        cc.caf_bins_dev(iq[0].data_ptr(), -100, 100)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            pkc, lagc = cc.caf_bins_dev(iq[0].data_ptr(), -5000, 5000)
            ts.append(time.perf_counter() - t0)
        cc.close()
        bi = int(np.argmax(pkc))
        pcx = Correlator(chips, fs=FS, Nint=0, device=local_rank, profile=True)
        pcx.caf_bins_dev(iq[0].data_ptr(), -100, 100)
        pcx.profile(reset=True)
        pcx.caf_bins_dev(iq[0].data_ptr(), -5000, 5000)
        cprof = pcx.profile()
        pcx.close()
        bpl = cprof["k_row_caf"]["units"] / cprof["k_row_caf"]["launches"] / N          # bins per launch
        cbytes = {"k_row_caf": lambda nb: nb * N * 16 + 8 * N * math.ceil(nb / 32),      # Y row in + bin buffer out per bin, code spectrum per 32-bin group
                  "k_col_inv_caf": lambda nb: nb * N * 8}                                 # bin buffer in
        ck = {k: v["ms_total"] / v["launches"] for k, v in cprof.items() if k in cbytes}
        cdom = max(ck, key=lambda k: cprof[k]["ms_total"])
        cach = cbytes[cdom](bpl) / (ck[cdom] * 1e-3) / 1e9
        sw = min(ts)
        out["caf_workload"] = {"workload": "BASELINE.json configs[2]: one 1-s window, +-5 kHz at 1 Hz = 10001 Doppler bins x 5e6 lags, HBM-resident",
                               "s_per_window": round(sw, 4), "Gsample_bins_per_s": round(10001 * N / sw / 1e9, 1),
                               "Msamples_per_s": round(N / sw / 1e6, 2), "peak_bin_hz": bi - 5000, "peak_lag": int(lagc[bi]),
                               "peak_exact": bool(bi - 5000 == 1781 and int(lagc[bi]) == delays[0]),
                               "bins_per_launch": int(round(bpl)),
                               "kernels": {k: {"avg_ms": round(v, 4), "us_per_bin": round(v * 1e3 / bpl, 3),
                                               "GB/s": round(cbytes[k](bpl) / (v * 1e-3) / 1e9, 1)} for k, v in ck.items()},
                               "roofline": {"bound": "hbm", "kernel": cdom, "achieved": round(cach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(cach / HBM_PEAK_GBS, 4), "traffic": None,
                                            "algorithmic_bytes_per_launch": int(cbytes[cdom](bpl)), "avg_ms": round(ck[cdom], 4),
                                            "note": "algorithmic bytes of the kernel's own interface; the Y rows and the bin buffer of a launch "
                                                    "are mostly served by L2 / Infinity Cache, so HBM traffic is below this figure"}}
        if pmc_caf and pmc_caf[0]:
            # HBM-true figures: the PMC bytes of one full launch of each kernel over the HIP-event duration of that launch
            tr = {"k_row_caf": pmc_caf[0]["k_rowd_caf"], "k_col_inv_caf": pmc_caf[0]["k_col_inv"]}
            rf = out["caf_workload"]["roofline"]
            for k in ck:
                out["caf_workload"]["kernels"][k]["hbm_traffic_bytes"] = int(tr[k])
                out["caf_workload"]["kernels"][k]["hbm_GB/s"] = round(tr[k] / (ck[k] * 1e-3) / 1e9, 1)
            pair_b, pair_ms = sum(tr[k] for k in ck), sum(ck.values())
            rf.update(traffic=int(tr[cdom]), achieved=round(tr[cdom] / (ck[cdom] * 1e-3) / 1e9, 1),
                      frac=round(tr[cdom] / (ck[cdom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), interface_GBs=round(cach, 1),
                      pair={"kernels": sorted(ck), "traffic": int(pair_b), "ms": round(pair_ms, 4), "GB/s": round(pair_b / (pair_ms * 1e-3) / 1e9, 1),
                            "frac": round(pair_b / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                      traffic_source=pmc_caf[1] + " (%.0f s)" % pmc_caf[2],
                      note="achieved / frac are HBM-true: PMC bytes of a full launch (read = 2 x FETCH_SIZE, write = WRITE_SIZE) over its duration; "
                           "interface_GBs prices the kernel's own interface bytes, most of which L2 / Infinity Cache serve")
        elif pmc_caf:
            out["caf_workload"]["roofline"]["traffic_live_error"] = pmc_caf[1]
    # --- BASELINE.json configs[4]: the wideband front end + four concurrent correlations, and the same chain in fp64
    if rank == 0 and world == 1 and not a.no_wideband and not a.no_roofline:
        del iq, gathered
        torch.cuda.empty_cache()
        try:
            out["wideband_workload"], out["f64_workload"] = wideband_legs(local_rank, a.wideband_seconds)
        except Exception as e:                               # the headline line is never lost to a side leg
            out["wideband_workload"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if strong is not None:
        out["strong_workload"] = strong
    if coll_info is not None:
        out["collective"] = coll_info
    else:
        out["numa"] = pin
    if world > 1:
        out["omitted_at_n_gt_1"] = {"cpu_baseline": "N=1 only (the bench contract: rank 0 at N=1)",
                                    "roofline.traffic": "the live rocprofv3 --pmc passes run at N=1 only; here from the committed profiles/pmc_traffic.json",
                                    "caf_workload": "N=1 only"}
    if cpu is not None:
        if "single_core" in cpu:
            n_cpu = cpu["single_core"]["windows"]
            names = [k for k in cpu if k == "single_core" or k.startswith("workers_") or k == "scipy_fft_workers_all"]
            best = max((cpu[k] for k in names if "value" in cpu[k]), key=lambda v: v["value"])
            which = [k for k in names if cpu.get(k) is best][0]
            out["cpu_baseline"] = {"value": best["value"], "unit": "Msamples/s", "cores": best["cores"], "kind": "port",
                                   "sample": f"{which}: {best['windows']} windows of the same workload ({best['windows'] * N} samples, "
                                             f"{best['seconds']} s of numpy fp64 processing() = oracle/twstft_oracle.py on the host's "
                                             f"{cpu['host_cores']} cores; one-off code-spectrum setup {cpu['setup_seconds']} s excluded)",
                                   "indice_matches_gpu": all(cpu["indices"][p] == int(arr[p].indice0) for p in range(min(n_cpu, nwin))),
                                   "variants": {k: cpu[k] for k in names}}
            if "corrections" in cpu and a.workload == "processing":
                # BASELINE.json's "delay err vs ref": the delay each window reports, (indice + correction) / ((2 Nint + 1) fs)
                # (godual_ranging.m:96), GPU against the fp64 oracle on the windows the CPU leg processed
                m = min(n_cpu, nwin)
                derr = [abs((int(arr[p].indice0) + arr[p].correction) - (cpu["indices"][p] + cpu["corrections"][p])) / (3.0 * FS) for p in range(m)]
                mrel = [abs(math.hypot(arr[p].xval[0], arr[p].xval[1]) - cpu["peak_mags"][p]) / cpu["peak_mags"][p] for p in range(m)]
                out["delay_err_ps"] = round(max(derr) * 1e12, 4)
                out["delay_err_vs_ref"] = {"max_ps": round(max(derr) * 1e12, 4), "mean_ps": round(sum(derr) / m * 1e12, 4), "windows": m,
                                           "integer_lag_differences": int(sum(1 for p in range(m) if cpu["indices"][p] != int(arr[p].indice0))),
                                           "peak_magnitude_max_rel_err": float("%.3g" % max(mrel)),
                                           "ref": "oracle/twstft_oracle.py processing() in fp64 (the reference's numpy path), same windows",
                                           "note": "one sample of the x3 grid = 66 667 ps; the gates are integer lag exact and |peak| within 1e-6"}
        else:
            out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out))
    cor.close()
    ex.close()                           # (barrier first: rank 0 may still have been profiling; leave together)


if __name__ == "__main__":
    main()
