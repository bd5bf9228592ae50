"""GPU (-m gpu): several GPUs from ONE host process — ``twx_multi_*`` (include/twstft_hip.h), the route a MATLAB / Octave / C
host has to more than one device (one context + one host thread per device, contiguous window blocks, one ncclAllGather).

A one-GPU box runs the threading and the ordering with a device list that repeats device 0 (blocks concatenated on the host)
and the RCCL calls with a world of one; wherever two GPUs are visible the same tests run the real collective.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd.correlator import ALL_CHANNELS, Correlator, band_godual
from amaranth_twstft_amd.multi import MultiCorrelator

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = 5e6
REC = C.sizeof(L.twx_result)


def _two_channel_capture(tmp_path, nwin, nchips=10000, seed=31):
    from tests.test_gpu_parity import _capture
    chips, raw = _capture(14, 43, nchips, nwin, seed=seed)
    path = tmp_path / "cap.bin"
    raw.tofile(path)
    return chips, raw, str(path)


@pytest.mark.parametrize("devices,nwin", [([0, 0, 0, 0], 13), ([0, 0, 0], 2), ([0] * 8, 600)])
def test_repeated_device_list_is_byte_identical_to_one_context(tmp_path, devices, nwin):
    """{0,0,0,0}: four contexts on four host threads share GPU 0, every one reads its own contiguous extent of the capture file;
    records byte-identical to one context's twx_process_file, in window order — ragged blocks (13 = 4+3+3+3), fewer windows
    than contexts (2 over 3), configs[3]'s 600 windows over 8.  File path and host-buffer path, one channel and all channels."""
    chips, raw, path = _two_channel_capture(tmp_path, nwin)
    n = 2 * len(chips)
    band = band_godual(FS, n)
    with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as one:
        ref_all = one.process_file(path, n_channels=2, channel=ALL_CHANNELS, band=band, raw_records=True)
        ref_ch1 = one.process_file(path, n_channels=2, channel=1, band=band, skip_samples=n, raw_records=True)
    assert ref_all.shape == (2 * nwin, REC)
    with MultiCorrelator(chips, devices, fs=FS, Nint=1, var_ddof=1) as m:
        info = m.info
        assert info.n_contexts == len(devices) and info.n_devices_distinct == 1 and info.rccl == 0
        got_all = m.process_file(path, n_channels=2, channel=ALL_CHANNELS, band=band, raw_records=True)
        assert got_all.tobytes() == ref_all.tobytes()
        got_ch1 = m.process_file(path, n_channels=2, channel=1, band=band, skip_samples=n, raw_records=True)     # skip + short job
        assert got_ch1.shape == (nwin - 1, REC) and got_ch1.tobytes() == ref_ch1.tobytes()
        if nwin <= 16:
            got_host = m.process(raw, n_channels=2, channel=ALL_CHANNELS, band=band, raw_records=True)          # capture in host memory
            assert got_host.tobytes() == ref_all.tobytes()
            dfs = np.linspace(-300.0, 300.0, nwin)
            with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as one:
                want = one.process(raw, n_channels=2, channel=0, df=dfs)
            got = m.process(raw, n_channels=2, channel=0, df=dfs)                                                # per-window df follows its window
            assert [g.indice for g in got] == [w.indice for w in want] and [g.df for g in got] == list(dfs)
        assert m.info.records_gathered >= 1


def test_device_resident_step_and_every_contexts_gathered_copy():
    """twx_multi_process_windows_dev: every context processes its OWN device-resident recording (the weak-scaling step of
    bench.py --single-process); all contexts end up with the same gathered buffer, block r holding recording r's records."""
    import torch
    from amaranth_twstft_amd import prn, synth
    chips = prn.lfsr_chips(14, 43, 10000)
    n, nwin, nctx = 20000, 5, 3
    dev = torch.device("cuda", 0)
    recs, delays = [], []
    for r in range(nctx):
        ps = [synth.SynthParams(delay_q8=(1000 * (r + 1) + 7 * w) * 256, fstep=synth.fstep_for_df(200.0 * r, FS), phi0=w, amp=300,
                                noise_gain=synth.noise_gain_for_sigma(300.0), seed=100 * r + w) for w in range(nwin)]
        raw = np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps])
        recs.append(torch.from_numpy(raw).to(dev))
        delays.append([1000 * (r + 1) + 7 * w for w in range(nwin)])
    band = band_godual(FS, n)
    with MultiCorrelator(chips, [0] * nctx, fs=FS, Nint=1) as m:
        got = m.process_dev([t.data_ptr() for t in recs], nwin, band=band)
        arr = (L.twx_result * (nctx * nwin)).from_buffer_copy(got.tobytes())
        for r in range(nctx):
            assert [int(arr[r * nwin + w].indice0) for w in range(nwin)] == [3 * d for d in delays[r]]
            assert m.fetch_gathered(r, nctx * nwin).tobytes() == got.tobytes()
    with Correlator(chips, fs=FS, Nint=1) as one:
        for r in range(nctx):
            want = one.process_dev(recs[r].data_ptr(), nwin, band=band)
            assert [w.indice for w in want] == [int(arr[r * nwin + w].indice0) for w in range(nwin)]
            assert [w.xval.real for w in want] == [arr[r * nwin + w].xval[0] for w in range(nwin)]


@pytest.mark.parametrize("nctx,nwin,rccl_one", [(3, 11, False), (8, 600, False), (5, 3, False), (1, 7, True)])
def test_one_recording_sharded_over_the_contexts(nctx, nwin, rccl_one):
    """twx_multi_process_recording_dev — BASELINE.json configs[3] as written (godual_ranging.m:75-102: ONE recording, consecutive
    windows): context i holds only ITS contiguous block on its device; the records come back in window order, byte-identical to one
    context processing the whole recording — ragged blocks (11 = 4+4+3), 600 over 8, fewer windows than contexts (3 over 5), and the
    RCCL world of one.  The exchange alone (twx_multi_exchange_only) re-delivers the same buffers."""
    import torch
    from amaranth_twstft_amd import prn, synth
    chips = prn.lfsr_chips(13, 27, 2500)
    n = 2 * len(chips)
    dev = torch.device("cuda", 0)
    ps = [synth.SynthParams(delay_q8=(300 + 5 * w) * 256, fstep=synth.fstep_for_df(150.0 + w, FS), phi0=w, amp=300,
                            noise_gain=synth.noise_gain_for_sigma(300.0), seed=700 + w) for w in range(nwin)]
    raw = np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps]).reshape(nwin, n, 2)
    whole = torch.from_numpy(raw).to(dev)
    band = band_godual(FS, n)
    with Correlator(chips, fs=FS, Nint=1) as one:
        want = one.process_dev(whole.data_ptr(), nwin, band=band)
    off = want[0].indice - 3 * 300                          # WindowResult.indice is Octave's (1-based)
    assert [w.indice for w in want] == [3 * (300 + 5 * w) + off for w in range(nwin)] and off in (0, 1)
    with MultiCorrelator(chips, [0] * nctx, fs=FS, Nint=1, rccl=rccl_one) as m:
        blocks = [m.block(nwin, r) for r in range(nctx)]
        assert sum(c for _, c in blocks) == nwin and blocks[0][0] == 0 and max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
        parts = [torch.from_numpy(raw[s0:s0 + c0].copy()).to(dev) if c0 else None for s0, c0 in blocks]      # every context sees ONLY its block
        got = m.process_recording_dev([t.data_ptr() if t is not None else 0 for t in parts], nwin, band=band)
        arr = (L.twx_result * nwin).from_buffer_copy(got.tobytes())
        assert [int(arr[w].indice0) + off for w in range(nwin)] == [w.indice for w in want]
        assert [(arr[w].xval[0], arr[w].xval[1], arr[w].df, arr[w].SNRr) for w in range(nwin)] == [(w.xval.real, w.xval.imag, w.df, w.SNRr) for w in want]
        cap = max(c for _, c in blocks)
        i = m.info
        assert i.records_gathered == nctx * cap and i.bytes_per_rank == cap * REC and i.rccl == (1 if rccl_one else 0)
        g0 = m.fetch_gathered(0, nctx * cap).tobytes()
        ms = m.exchange_only(cap)
        assert ms >= 0 and m.fetch_gathered(nctx - 1, nctx * cap).tobytes() == g0            # the exchange alone delivers the same buffers again
        # supplied carriers follow their windows across the block boundaries
        dfs = np.linspace(140.0, 160.0 + nwin, nwin)
        got2 = m.process_recording_dev([t.data_ptr() if t is not None else 0 for t in parts], nwin, df=dfs)
        arr2 = (L.twx_result * nwin).from_buffer_copy(got2.tobytes())
        assert [arr2[w].df for w in range(nwin)] == list(dfs)
        with pytest.raises(L.TwxError):
            m.exchange_only(10 ** 9)


def test_bench_single_process_strong_leg():
    """`bench.py --gpus 4 --single-process`: besides the weak headline the line times configs[3] as written — ONE recording of `--windows`
    windows sharded over the four contexts — with the exchange isolated; every context's block carries the generator's lags."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--single-process", "--steps", "2", "--warmup", "1",
                          "--windows", "9"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    s = j["strong_workload"]
    assert j["scaling"] == "weak" and s["scaling"] == "strong" and s["windows_total"] == 9 and s["windows_per_rank"] == [3, 2, 2, 2]
    assert s["ranks_with_exact_lags"] == 4 and s["all_ranks_agree"] and s["value"] > 0 and s["gather_ms"] >= 0 and s["ms_per_step"] > 0


def test_rccl_world_of_one_runs_the_collective_calls(tmp_path):
    """One device, RCCL forced on: ncclCommInitAll + ncclGroupStart / ncclAllGather / ncclGroupEnd of a world of one — the calls
    of the N > 1 path, bound from librccl.so.1 at run time, on a box with a single GPU.  Same records as without."""
    chips, raw, path = _two_channel_capture(tmp_path, 7)
    band = band_godual(FS, 2 * len(chips))
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=False) as a:
        ref = a.process_file(path, n_channels=2, channel=0, band=band, raw_records=True)
        assert a.info.rccl == 0
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=True) as b:
        got = b.process_file(path, n_channels=2, channel=0, band=band, raw_records=True)
        i = b.info
        assert i.rccl == 1 and i.rccl_version > 0 and i.records_gathered == 7 and i.bytes_per_rank == 7 * REC
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("inject,stage", [("init", 1), ("init_hang", 1), ("gather", 2), ("gather_timeout", 2)])
def test_rccl_failures_fall_back_to_host_concatenation(tmp_path, monkeypatch, inject, stage):
    """The exchange must never lose the job: ncclCommInitAll failing or not returning (stage 1), a collective failing or never
    completing (stage 2) — injected, a one-GPU box has no other way in — leave the driver on host-side concatenation, flagged in
    twx_multi_info with the reason; records byte-identical to the run without RCCL, file path and device-resident path."""
    import torch
    chips, raw, path = _two_channel_capture(tmp_path, 7)
    n = 2 * len(chips)
    band = band_godual(FS, n)
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=False) as a:
        ref = a.process_file(path, n_channels=2, channel=0, band=band, raw_records=True)
        assert a.info.rccl == 0 and a.info.rccl_fallback == 0 and a.info.rccl_error == b""
    monkeypatch.setenv("TWX_MULTI_INJECT", inject)
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=True) as b:
        i = b.info
        if stage == 1:
            assert i.rccl == 0 and i.rccl_fallback == 1 and b"ncclCommInitAll" in i.rccl_error and b"injected" in i.rccl_error
        else:
            assert i.rccl == 1 and i.rccl_fallback == 0
        got = b.process_file(path, n_channels=2, channel=0, band=band, raw_records=True)
        i = b.info
        assert i.rccl == 0 and i.rccl_fallback == stage and i.records_gathered == 7
        assert (b"ncclAllGather" in i.rccl_error) == (stage == 2), i.rccl_error
        assert got.tobytes() == ref.tobytes()
        again = b.process_file(path, n_channels=2, channel=0, band=band, raw_records=True)        # stays on the host path, quietly
        assert again.tobytes() == ref.tobytes() and b.info.rccl_fallback == stage
    # the device-resident step (bench.py --single-process): the fall-back happens inside the very call whose collective failed
    dev_raw = torch.from_numpy(raw.reshape(-1, 4)[:, :2].copy().reshape(-1)).to("cuda:0")
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=True) as c:
        got_dev = c.process_dev([dev_raw.data_ptr()], 7, band=band)
        assert c.info.rccl == 0 and c.info.rccl_fallback == stage
        assert c.fetch_gathered(0, 7).tobytes() == got_dev.tobytes()
    monkeypatch.delenv("TWX_MULTI_INJECT")
    with MultiCorrelator(chips, [0], fs=FS, Nint=1, rccl=True) as d:
        want_dev = d.process_dev([dev_raw.data_ptr()], 7, band=band)
        assert d.info.rccl == 1 and d.info.rccl_fallback == 0
    assert got_dev.tobytes() == want_dev.tobytes()


def test_workers_are_placed_next_to_their_device():
    """NUMA placement: twx_device_affinity answers from /sys/bus/pci/devices/<bus id>/, the twx_multi workers bind themselves to those
    CPUs where the platform names a node (threads_pinned), and a rank's own thread can do the same (twx_pin_thread_to_device)."""
    from amaranth_twstft_amd import prn
    lib = L.load()
    node, buf = C.c_int32(-7), C.create_string_buffer(512)
    assert lib.twx_device_affinity(0, C.byref(node), buf, 512) == 0 and node.value >= -1
    assert lib.twx_device_affinity(99, C.byref(node), buf, 512) != 0
    before = os.sched_getaffinity(0)
    try:
        n2, k = C.c_int32(-7), C.c_int32(-7)
        assert lib.twx_pin_thread_to_device(0, C.byref(n2), C.byref(k)) == 0 and n2.value == node.value
        if node.value >= 0 and buf.value:
            assert k.value >= 1 and len(os.sched_getaffinity(0)) == k.value
        else:
            assert k.value == 0 and os.sched_getaffinity(0) == before              # the platform does not say: nothing changed
    finally:
        os.sched_setaffinity(0, before)
    chips = prn.lfsr_chips(13, 27, 5000)
    with MultiCorrelator(chips, [0, 0, 0], fs=FS, Nint=1) as m:
        i = m.info
        assert [i.numa_node[r] for r in range(3)] == [node.value] * 3 and i.numa_node[3] == -1
        assert i.threads_pinned == (3 if node.value >= 0 and buf.value else 0)
    assert os.sched_getaffinity(0) == before                                       # the caller's own thread is never touched by twx_multi


def test_two_real_devices_gather_over_rccl(tmp_path):
    """Two distinct devices: the record exchange is one ncclAllGather over xGMI (skipped on a one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL has one rank per device")
    chips, raw, path = _two_channel_capture(tmp_path, 13)
    band = band_godual(FS, 2 * len(chips))
    with Correlator(chips, fs=FS, Nint=1) as one:
        ref = one.process_file(path, n_channels=2, channel=ALL_CHANNELS, band=band, raw_records=True)
    with MultiCorrelator(chips, [0, 1], fs=FS, Nint=1) as m:
        got = m.process_file(path, n_channels=2, channel=ALL_CHANNELS, band=band, raw_records=True)
        assert m.info.rccl == 1 and m.info.n_devices_distinct == 2
    assert got.tobytes() == ref.tobytes()


def test_errors_come_back_as_status_codes(tmp_path):
    chips, raw, path = _two_channel_capture(tmp_path, 3)
    lib = L.load()
    h = C.c_void_p()
    cfg = L.twx_config()
    cfg.fs, cfg.sps, cfg.nint, cfg.lfsr_bitlen, cfg.lfsr_taps, cfg.n_chips = FS, 2, 1, 14, 43, 10000
    devs = (C.c_int32 * 2)(0, 99)
    assert lib.twx_multi_create(C.byref(cfg), C.cast(devs, C.c_void_p), 2, 0, C.byref(h)) == -1 and not h.value
    assert b"does not exist" in lib.twx_multi_last_error(None)
    with MultiCorrelator(chips, [0, 0], fs=FS, Nint=1) as m:
        with pytest.raises(L.TwxError, match="cannot open"):
            m.process_file(str(tmp_path / "nope.bin"), n_channels=2, channel=0, band=(0, 10), max_windows=1)
        out = m.process_file(path, n_channels=2, channel=0, band=band_godual(FS, 20000), max_windows=0)
        assert out == []


def test_script_level_job_single_process(tmp_path):
    """python -m amaranth_twstft_amd.godual_ranging --gpus 4 --single-process: the file-in / delay-out job of configs[3] through
    twx_multi (no torch.distributed, no child ranks) writes the .mat / TSV a one-GPU run writes, byte for byte."""
    from tests.test_gpu_configs import _write_capture
    _write_capture(tmp_path, 10000, 13, seed=77)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "amaranth_twstft_amd.godual_ranging", "--datalocation", str(tmp_path), "--codelocation", str(tmp_path / "codes")]
    one = subprocess.run(base, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    mat = tmp_path / "1670074501.mat"
    ref_bytes = mat.read_bytes()[128:]
    mat.unlink()
    many = subprocess.run(base + ["--gpus", "4", "--single-process"], capture_output=True, text=True, env=env, timeout=900)
    assert many.returncode == 0, many.stdout[-2000:] + many.stderr[-3000:]
    assert mat.read_bytes()[128:] == ref_bytes
    rows = lambda s: [l for l in s.splitlines() if l[:1].isdigit() and "\t" in l]
    assert rows(many.stdout) == rows(one.stdout) and len(rows(one.stdout)) == 13


def test_mex_gateway_ngpu_argument_and_file_form(tmp_path):
    """twstft_processing_mex executed on the GPU (functional fake mex.h): call form B with a trailing ngpu = 3 and call form C
    ('file', path, ...) with ngpu 1 and 4, skip and max_windows — every output equal to the one-GPU raw form, value for value."""
    from tests.test_abi_and_host import build_mex_harness
    from tests.test_gpu_configs import _read_mex_outputs
    from tests.test_gpu_parity import _capture
    exe = build_mex_harness(ROOT, tmp_path)
    nchips, n, nwin = 10000, 20000, 7
    chips, raw = _capture(14, 43, nchips, nwin, seed=64)
    raw.tofile(tmp_path / "cap.bin")
    chips.tofile(tmp_path / "chips.bin")
    band = band_godual(FS, n)
    common = [str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"), "2"]
    kk = [str(band[0] + 1), str(band[1] + 1)]

    def run(mode, chan, *tail):
        r = subprocess.run([str(exe), mode, *common, str(chan), *kk, "5e6", "1", *tail], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return _read_mex_outputs(tmp_path / "out.bin")

    ref = run("raw", 0)
    assert all(x.shape == (2, nwin) for x in ref)
    for got in (run("raw", 0, "ngpu=3"), run("file", 0), run("file", 0, "ngpu=4"), run("file", 0, "godual", "ngpu=2")):
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    part = run("file", 2, "ngpu=3", "skip=%d" % (2 * n), "max=4")              # channel 2, windows 2..5
    assert all(x.shape == (1, 4) for x in part)
    assert all(np.array_equal(a[0], b[1, 2:6]) for a, b in zip(part, ref))


def test_bench_single_process_line():
    """`python bench.py --gpus 4 --single-process`: four contexts driven from the one process (sharing GPU 0 on a one-GPU box),
    the line carries the same `collective` object as the one-rank-per-GPU form and every context's gathered copy was checked."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--single-process", "--steps", "2", "--warmup", "1",
                          "--windows", "9"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = j["collective"]
    assert j["n_gpus"] == 4 and j["integer_lag_exact"] and j["value"] > 0 and j["config"]["launch"] == "single process"
    assert c["world"] == 4 and c["records"] == 36 and c["ranks_with_exact_lags"] == 4 and c["gathered_lag_exact"] and c["all_ranks_agree"]


def test_randomised_partitions_equal_one_context(tmp_path):
    """Random jobs over random device lists (device 0 repeated 1..7 times): window count from 0 upward, one or two channels, one
    channel / all channels, band or supplied carrier, file skip and window limit, host-buffer and file entry — the gathered records
    byte-identical to ONE context's, whatever the partition (blocks of 0 windows, fewer windows than contexts, ragged blocks)."""
    from tests.test_gpu_parity import _capture
    rng = np.random.default_rng(99 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    chips, raw_all = _capture(13, 27, 5000, 23, seed=5)
    n = 2 * len(chips)
    band = band_godual(FS, n)
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "10"))
    with Correlator(chips, fs=FS, Nint=1) as one:
        for it in range(ncomb):
            ndev = int(rng.integers(1, 8))
            nwin = int(rng.choice([0, 1, 2, ndev - 1, ndev, ndev + 1, rng.integers(0, 24)]))
            nwin = max(0, min(23, nwin))
            channel = int(rng.integers(-1, 2))
            use_band = bool(rng.integers(0, 2))
            skip = int(rng.integers(0, 3)) * n if nwin > 2 else 0
            maxw = int(rng.integers(1, nwin + 2)) if rng.integers(0, 2) else None
            raw = raw_all[: nwin * n]
            path = tmp_path / f"cap{it}.bin"
            raw.tofile(path)
            tag = f"combination {it}: {ndev} contexts, {nwin} windows, channel {channel}, {'band' if use_band else 'df'}, skip {skip // n}, max {maxw}"
            kw = dict(band=band) if use_band else dict(df=123.5)
            with MultiCorrelator(chips, [0] * ndev, fs=FS, Nint=1) as m:
                ref = one.process_file(str(path), n_channels=2, channel=channel, skip_samples=skip, max_windows=maxw, raw_records=True, **kw)
                got = m.process_file(str(path), n_channels=2, channel=channel, skip_samples=skip, max_windows=maxw, raw_records=True, **kw)
                assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), tag
                if nwin:
                    kw2 = dict(band=band) if use_band else dict(df=np.linspace(-50.0, 50.0, nwin * (2 if channel < 0 else 1)).reshape((nwin, 2) if channel < 0 else (nwin,)))
                    r2 = one.process(raw, n_channels=2, channel=channel, **kw2)
                    g2 = m.process(raw, n_channels=2, channel=channel, **kw2)
                    flat = lambda r: [x for c in sorted(r) for x in r[c]] if isinstance(r, dict) else r
                    assert [(a.indice, a.xval, a.df, a.SNRr) for a in flat(g2)] == [(a.indice, a.xval, a.df, a.SNRr) for a in flat(r2)], tag + " (host buffer)"
