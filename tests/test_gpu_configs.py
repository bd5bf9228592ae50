"""GPU (-m gpu): BASELINE.json configs[2], [3] and [4] at their STATED sizes, the multi-rank path with the real
correlator, and the single-slot ingest pipeline.

Gates as everywhere (north_star): integer lag bit-exact, |peak| within 1e-6 relative (fp32 vs the fp64 oracle / fp64 context).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import dist as D
from amaranth_twstft_amd import frontend, prn, synth
from amaranth_twstft_amd.correlator import ALL_CHANNELS, Correlator, band_godual
from oracle import twstft_oracle as orc
from tests.helpers import chips_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = 5e6
MAG_TOL = 1e-6
NCHIPS, N = 2_500_000, 5_000_000


def _params(p):
    return np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)


def _synth_dev(out, n, chips_dev, nchips, sps, chans, n0=0):
    """Device generator (bit-identical to synth.synth_capture) into the torch int16 tensor ``out`` [n, 2*len(chans)]."""
    lib = L.load()
    params = np.concatenate([_params(p) for p in chans])
    L.check(lib.twx_synth_capture_dev(out.data_ptr(), n, n0, chips_dev.data_ptr(), nchips, sps, len(chans),
                                      params.ctypes.data_as(C.c_void_p), None))


# --------------------------------------------------------------------------------------------------------------
# configs[2]: 1-s capture, full delay x Doppler surface, +-5 kHz at 1 Hz (10 001 bins of fs/N = 1 Hz), N = 5e6
# --------------------------------------------------------------------------------------------------------------
def test_config2_full_caf_grid_at_size():
    import torch
    chips = chips_for(22, 3, NCHIPS)
    dev = torch.device("cuda", 0)
    p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(1780.75, FS), phi0=0, amp=200,
                          noise_gain=synth.noise_gain_for_sigma(400.0), seed=7)
    iq = torch.empty((N, 2), dtype=torch.int16, device=dev)
    _synth_dev(iq, N, torch.from_numpy(chips).to(dev), NCHIPS, 2, [p])
    torch.cuda.synchronize()
    raw = iq.cpu().numpy()
    del iq
    k_lo, k_hi = -5000, 5000
    with Correlator(chips, fs=FS, Nint=0) as cor:
        pk, lag = cor.caf_bins(raw, k_lo, k_hi)
        # second device route: the full processing(d,df) chain once per trial offset f = kappa*fs/N
        res = cor.caf_freqs(raw, np.arange(k_lo, k_hi + 1, dtype=np.float64) * (FS / N))
    assert pk.shape == (10001,)
    best = int(np.argmax(pk))
    assert best + k_lo == 1781 and lag[best] == 1311765            # df = 1780.75 Hz -> nearest 1-Hz bin; delay of the generator
    assert pk[best] > 5 * np.median(pk)
    # every bin against the second route (full chain per trial offset) and against the SAME surface computed in fp64 on the
    # device: peaks within the gate, and every one of the 10 001 lags decided — where two routes disagree the fp64 map of that
    # bin must show the two lags tied to within fp32 resolution (no allowance otherwise)
    lag2 = np.array([r.indice for r in res])
    pk2 = np.array([abs(r.xval) for r in res])
    assert np.abs(pk - pk2).max() <= 2 * MAG_TOL * pk.max()
    with Correlator(chips, fs=FS, Nint=0, precision="f64") as c64:
        pk64, lag64 = c64.caf_bins(raw, k_lo, k_hi)
        assert np.abs(pk - pk64).max() <= MAG_TOL * pk64.max()
        for i in np.nonzero((lag != lag64) | (lag2 != lag64))[0]:
            m = np.abs(c64.xcorr_map(raw, float(i + k_lo) * (FS / N)))
            assert int(m.argmax()) == lag64[i]
            for cand in (lag[i], lag2[i]):
                assert m[cand] >= m[lag64[i]] * (1 - MAG_TOL), (i, cand)             # a tie at fp32 resolution, nothing else
    # 256 bins spread over the grid (+ the peak bin, its neighbours, bin 0 and both edges) against the ORACLE: bit-exact lags,
    # for the fp32 and the fp64 surface (host FFTs on a thread pool: numpy releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    fcode = orc.make_fcode(orc.make_code(chips, 2))
    Y = np.fft.fft(d)
    ks = sorted(set(list(np.linspace(k_lo, k_hi, 256).astype(int)) + [1781, 1780, 1782, 0]))

    def one(kk):
        m = np.abs(np.fft.ifft(np.roll(Y, -kk) * fcode))
        j = int(m.argmax())
        return kk, j, float(m[j])

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for kk, j, mj in ex.map(one, ks):
            assert lag[kk - k_lo] == j and lag64[kk - k_lo] == j, kk
            assert abs(pk[kk - k_lo] - mj) <= MAG_TOL * mj and abs(pk64[kk - k_lo] - mj) <= 1e-9 * mj, kk
    assert len(ks) >= 256


# --------------------------------------------------------------------------------------------------------------
# configs[3]: 600 x 1-s windows, 2 channels, through shard -> process -> gather
# --------------------------------------------------------------------------------------------------------------
def _c4_channels(p):
    import math
    delay = 1311765 - int(round(0.025 * p))
    df = 1780.75 + 0.05 * math.sin(2 * math.pi * p / 600)
    ch1 = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(df, FS), phi0=(p * 2654435761) & 0xFFFFFFFF, amp=200,
                            noise_gain=synth.noise_gain_for_sigma(400.0), seed=1000 + p, stream=0)
    ch2 = synth.SynthParams(delay_q8=3626553 * 256, fstep=0, phi0=12345, amp=3000, noise_gain=synth.noise_gain_for_sigma(100.0),
                            seed=1000 + p, stream=1)                                              # loop-back channel
    return delay, [ch1, ch2]


def test_config3_600_windows_two_channels_one_gpu():
    """The whole C4 recording (600 windows x 2 channels, 24 GB of int16 resident in HBM) on ONE GPU: every lag equals
    3*delay_p of the generator, df follows the generator's drift, results go through dist.shard/gather (world 1)."""
    import torch
    nwin = 600
    chips = chips_for(22, 3, NCHIPS)
    dev = torch.device("cuda", 0)
    chips_dev = torch.from_numpy(chips).to(dev)
    iq = torch.empty((nwin, N, 4), dtype=torch.int16, device=dev)
    delays = []
    for p in range(nwin):
        delay, chans = _c4_channels(p)
        delays.append(delay)
        _synth_dev(iq[p], N, chips_dev, NCHIPS, 2, chans)
    torch.cuda.synchronize()
    start, stop = D.shard_windows(nwin, 0, 1)
    assert (start, stop) == (0, 600) and D.shard_windows(nwin, 3, 8) == (225, 300)
    res = torch.zeros((nwin * 2, D.RESULT_BYTES), dtype=torch.uint8, device=dev)
    lib = L.load()
    with Correlator(chips, fs=FS, Nint=1) as cor:
        band = L.twx_band(*band_godual(FS, N))
        L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 2, ALL_CHANNELS, C.byref(band), None, res.data_ptr()), cor._h)
        cor.synchronize()
    allb = D.gather_results(res, nwin, 0, 1, per_window=2)
    recs = D.results_from_bytes(allb)
    r1, r2 = recs[0::2], recs[1::2]
    assert len(r1) == len(r2) == nwin
    import math
    for p in range(nwin):
        assert r1[p].indice == 3 * delays[p], p
        assert r2[p].indice == 3 * 3626553, p
        df_true = synth.df_of_fstep(synth.fstep_for_df(1780.75 + 0.05 * math.sin(2 * math.pi * p / 600), FS), FS)
        # 0.5-Hz bins (fs/N/2) plus the +0.25 Hz the reference's linspace(-fs/2,fs/2,N) axis adds near DC (godual_ranging.m:73)
        assert abs(r1[p].df - df_true) <= 0.6
        assert abs(r2[p].df) <= 0.6
        assert abs(r1[p].correction) < 0.1 and abs(r2[p].correction) < 0.1
    # the same windows in 8 shards of 75, each as its own call on its own extent (what the 8 ranks do): identical records
    with Correlator(chips, fs=FS, Nint=1) as cor:
        for r in (0, 3, 7):
            s, e = D.shard_windows(nwin, r, 8)
            assert e - s == 75
            part = torch.zeros(((e - s) * 2, D.RESULT_BYTES), dtype=torch.uint8, device=dev)
            L.check(lib.twx_process_windows_dev(cor._h, iq[s].data_ptr(), e - s, 2, ALL_CHANNELS, C.byref(band), None, part.data_ptr()), cor._h)
            cor.synchronize()
            assert torch.equal(part, res[2 * s:2 * e])


# --------------------------------------------------------------------------------------------------------------
# configs[4]: two stations, 4 concurrent correlations, 70 Msps wideband capture, fp64 vs fp32 tolerance check
# --------------------------------------------------------------------------------------------------------------
def test_config4_two_station_wideband_chain_at_size():
    """Each station records, at 70 Msps, its own (local) code near 0 Hz plus the other station's (remote) code near
    +-50 kHz carrier offset, i.e. inside the +-(80..120) kHz band of the squared spectrum (godual_ranging.m:83-89).
    Device chain: FIR low-pass + decimate by 14 (twx_fir_decimate_dev) -> 1-s window of N = 5e6 at 5 Msps ->
    processing(d,k) in fp32 AND fp64 for the four correlations oplo/opre/ltlo/ltre of acquisition/go_1s.m:88,120,147,171
    (codes: taps 57 = OP, taps 3 = LTFB).  Gates: same lag fp32 = fp64 = oracle, |xval| within 1e-6 relative."""
    import torch
    dev = torch.device("cuda", 0)
    fs_in, dec, sps_in = 70e6, 14, 28
    taps = frontend.lowpass_taps(fs_in, 2.1e6, 0.4e6)
    ntaps = taps.size
    n_in = (N - 1) * dec + ntaps
    codes = {"OP": chips_for(22, 57, NCHIPS), "LTFB": chips_for(22, 3, NCHIPS)}
    cdev = {k: torch.from_numpy(v).to(dev) for k, v in codes.items()}
    half = (ntaps - 1) // 2
    # (station, local delay [70 Msps samples], remote delay, remote carrier offset, OP flag of the remote band)
    stations = [("OP", "LTFB", 18_364_717, 50_772_133, +50_000.0, 0), ("LTFB", "OP", 41_000_003, 9_123_457, -50_000.0, 1)]
    narrow = {}
    with Correlator(codes["OP"], fs=FS, Nint=1) as c_op32, Correlator(codes["OP"], fs=FS, Nint=1, precision="f64") as c_op64, \
            Correlator(codes["LTFB"], fs=FS, Nint=1) as c_lt32, Correlator(codes["LTFB"], fs=FS, Nint=1, precision="f64") as c_lt64:
        ctx = {"OP": (c_op32, c_op64), "LTFB": (c_lt32, c_lt64)}
        results = {}
        for st, other, d_loc, d_rem, f_rem, op_flag in stations:
            wide = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
            tmp = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
            ploc = synth.SynthParams(delay_q8=d_loc * 256, fstep=synth.fstep_for_df(3.25, fs_in), phi0=99, amp=2500,
                                     noise_gain=synth.noise_gain_for_sigma(2500.0), seed=401, stream=len(results))
            prem = synth.SynthParams(delay_q8=d_rem * 256, fstep=synth.fstep_for_df(f_rem, fs_in), phi0=7, amp=1200,
                                     noise_gain=0, seed=402, stream=len(results))
            _synth_dev(wide, n_in, cdev[st], NCHIPS, sps_in, [ploc])
            _synth_dev(tmp, n_in, cdev[other], NCHIPS, sps_in, [prem])
            torch.cuda.synchronize()
            wide = (wide.to(torch.int32) + tmp.to(torch.int32)).clamp_(-32768, 32767).to(torch.int16).contiguous()
            del tmp
            torch.cuda.synchronize()            # torch's stream wrote `wide`; the library's streams do not wait for it by themselves
            nar = torch.empty((N, 2), dtype=torch.int16, device=dev)
            narf = torch.empty((N, 2), dtype=torch.float32, device=dev)
            c32 = ctx[st][0]
            nout = c32.fir_decimate_dev(wide.data_ptr(), n_in, taps, dec, out_i16_dev=nar.data_ptr(), out_f32_dev=narf.data_ptr())
            assert nout == N
            c32.synchronize()
            # FIR parity on 4096 random outputs against the fp64 direct sum (the oracle's definition, orc.fir_decimate)
            rng = np.random.default_rng(3)
            ms = np.sort(rng.choice(N, 4096, replace=False))
            idx = torch.from_numpy((ms[:, None] * dec + np.arange(ntaps)[None, :]).reshape(-1)).to(dev)
            seg = wide[idx].cpu().numpy().astype(np.float64).reshape(4096, ntaps, 2)
            ref = (seg * taps.astype(np.float64)[None, :, None]).sum(axis=1)
            got = narf[torch.from_numpy(ms).to(dev)].cpu().numpy().astype(np.float64)
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-3
            g16 = nar[torch.from_numpy(ms).to(dev)].cpu().numpy()
            assert np.abs(g16 - np.rint(ref)).max() <= 1
            narrow[st] = nar
            del wide, narf
            # local correlation (own code, +-20 kHz band) and remote correlation (other code, remote band)
            for name, code_key, band in ((st + "lo", st, band_godual(FS, N)), (st + "re", other, band_godual(FS, N, remote=1, OP=op_flag))):
                g32 = ctx[code_key][0].process_dev(nar.data_ptr(), 1, band=band)[0]
                g64 = ctx[code_key][1].process_dev(nar.data_ptr(), 1, band=band)[0]
                results[name] = (g32, g64, code_key, band)
            # expected lags: y[m] is centred on input sample m*dec + (ntaps-1)/2
            for name, d70 in ((st + "lo", d_loc), (st + "re", d_rem)):
                expect = (d70 - half) / dec
                assert abs(results[name][0].indice / 3.0 - expect) < 1.0, name
        assert sorted(results) == ["LTFBlo", "LTFBre", "OPlo", "OPre"]
        # ... and the four IN FLIGHT TOGETHER: four contexts (own streams each), four launches back to back, one wait at the end —
        # record for record what the one-after-another calls above returned
        lib = L.load()
        with Correlator(codes["OP"], fs=FS, Nint=1) as x_op, Correlator(codes["LTFB"], fs=FS, Nint=1) as x_lt:
            four = {"OPlo": c_op32, "OPre": c_lt32, "LTFBlo": x_lt, "LTFBre": x_op}          # a context per correlation
            outs = {k: torch.zeros(C.sizeof(L.twx_result), dtype=torch.uint8, device=dev) for k in four}
            for name, cx in four.items():
                b = L.twx_band(*results[name][3])
                L.check(lib.twx_process_windows_dev(cx._h, narrow[name[:-2]].data_ptr(), 1, 1, 0, C.byref(b), None, outs[name].data_ptr()), cx._h)
            for cx in four.values():
                cx.synchronize()
            for name in four:
                r = L.twx_result.from_buffer_copy(outs[name].cpu().numpy().tobytes())
                g32 = results[name][0]
                assert int(r.indice0) == g32.indice and complex(*r.xval) == g32.xval and r.df == g32.df and r.correction == g32.correction, name
    # oracle on the same decimated int16 samples, all four correlations
    for name, (g32, g64, code_key, band) in results.items():
        st = name[:-2]
        raw = narrow[st].cpu().numpy()
        d = orc.deinterleave(raw, 1, 0)
        d = d - d.mean()
        code = orc.make_code(codes[code_key], 2)
        freq = orc.freq_axis(FS, N)
        k = np.arange(band[0], band[1] + 1)
        o = orc.processing(d, k, freq, np.arange(N) / FS, orc.make_fcode(code), code, Nint=1, fs=FS)
        assert g32.indice == g64.indice == o["indice"], name                       # delta indice = 0
        assert abs(g32.df - o["df"]) <= 1e-9 and abs(g64.df - o["df"]) <= 1e-9, name
        assert abs(abs(g32.xval) - abs(g64.xval)) <= MAG_TOL * abs(g64.xval), name  # fp32 vs fp64 peak magnitude
        assert abs(abs(g32.xval) - abs(o["xval"])) <= MAG_TOL * abs(o["xval"]), name
        assert abs(abs(g64.xval) - abs(o["xval"])) <= 1e-9 * abs(o["xval"]), name
        assert abs(g32.correction - o["correction"]) <= 2e-4 and abs(g64.correction - o["correction"]) <= 1e-7, name
        if name.endswith("re"):
            assert 40_000.0 < abs(o["df"]) < 60_000.0


def test_config4_session_steps_enqueued_behind_each_other():
    """amaranth_twstft_amd/wideband.py: five steps submitted back to back (double-buffered decimated captures and records, the
    front end of step i+1 sharing the GPU with the correlations of step i), records fetched one step behind.  The captures of
    consecutive steps DIFFER (other delays), so a record computed from a buffer that was overwritten too early, or read too
    early, cannot equal the reference: the same step done alone on fresh contexts with a wait after every call."""
    import torch
    from amaranth_twstft_amd.wideband import WidebandSession, godual_plan
    dev = torch.device("cuda", 0)
    fs_in, dec, sps_in = 70e6, 14, 28
    taps = frontend.lowpass_taps(fs_in, 2.1e6, 0.4e6)
    codes = {"OP": chips_for(22, 57, NCHIPS), "LTFB": chips_for(22, 3, NCHIPS)}
    cdev = {k: torch.from_numpy(v).to(dev) for k, v in codes.items()}
    n_in = (N - 1) * dec + taps.size
    half = (taps.size - 1) // 2

    def capture(st, other, d_loc, d_rem, f_rem, stream):
        wide = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        tmp = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        _synth_dev(wide, n_in, cdev[st], NCHIPS, sps_in, [synth.SynthParams(delay_q8=d_loc * 256, fstep=synth.fstep_for_df(3.25, fs_in), phi0=99, amp=2500,
                                                                          noise_gain=synth.noise_gain_for_sigma(2500.0), seed=411, stream=stream)])
        _synth_dev(tmp, n_in, cdev[other], NCHIPS, sps_in, [synth.SynthParams(delay_q8=d_rem * 256, fstep=synth.fstep_for_df(f_rem, fs_in), phi0=7, amp=1200,
                                                                            noise_gain=0, seed=412, stream=stream)])
        torch.cuda.synchronize()
        wide = (wide.to(torch.int32) + tmp.to(torch.int32)).clamp_(-32768, 32767).to(torch.int16).contiguous()
        torch.cuda.synchronize()
        return wide

    # two different capture sets, used alternately: A B A B A
    delays = [{"OP": (18_364_717, 50_772_133), "LTFB": (41_000_003, 9_123_457)}, {"OP": (3_141_593, 27_182_818), "LTFB": (60_221_409, 16_180_339)}]
    sets = [{"OP": capture("OP", "LTFB", *d["OP"], +50_000.0, 2 * i), "LTFB": capture("LTFB", "OP", *d["LTFB"], -50_000.0, 2 * i + 1)} for i, d in enumerate(delays)]
    plan = godual_plan(("OP", "LTFB"), FS, N)
    key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df, r.correction, r.xvalm1[0], r.xvalp1[0], r.SNRr, r.SNRi)
    # reference: one call at a time, a wait after each
    ref = []
    for cs in sets:
        out = {}
        nar = torch.empty((N, 2), dtype=torch.int16, device=dev)
        for st in ("OP", "LTFB"):
            for name, (cst, code_st, band) in plan.items():
                if cst != st:
                    continue
                with Correlator(codes[code_st], fs=FS, Nint=1) as c:
                    assert c.fir_decimate_dev(cs[st].data_ptr(), n_in, taps, dec, out_i16_dev=nar.data_ptr()) == N
                    c.synchronize()
                    b = L.twx_band(*band)
                    o = torch.zeros(C.sizeof(L.twx_result), dtype=torch.uint8, device=dev)
                    L.check(c._lib.twx_process_windows_dev(c._h, nar.data_ptr(), 1, 1, 0, C.byref(b), None, o.data_ptr()), c._h)
                    c.synchronize()
                    out[name] = key(L.twx_result.from_buffer_copy(o.cpu().numpy().tobytes()))
        ref.append(out)
    for i, d in enumerate(delays):               # the reference itself sits where the generator put the codes
        for st in ("OP", "LTFB"):
            for suffix, d70 in zip(("lo", "re"), d[st]):
                assert abs(ref[i][st + suffix][0] / 3.0 - (d70 - half) / dec) < 1.0
    assert ref[0] != ref[1]
    with WidebandSession(codes, taps, dec, fs=FS, windows=1, plan=plan) as sess:
        got = {}
        for i in range(5):
            step = sess.submit({st: t.data_ptr() for st, t in sets[i % 2].items()})
            assert step == i
            if i:
                got[i - 1] = sess.fetch(i - 1)
        got[4] = sess.fetch(4)
        with pytest.raises(ValueError):
            sess.fetch(2)                           # overwritten two steps ago
        for i in range(5):
            for name in plan:
                assert key(got[i][name][0]) == ref[i % 2][name], (i, name)
        # a capture too short for the step's windows is refused before anything is enqueued
        with pytest.raises(ValueError):
            sess.submit({st: t.data_ptr() for st, t in sets[0].items()}, n_in=n_in - dec)
        assert sess.step == 5


@pytest.mark.parametrize("depth", [1, 3])
def test_wideband_session_short_codes_any_depth(depth):
    """The session with 5 000-chip codes (N = 10 000, plan plug-in 25 x 400), three windows per step, a 57-tap front end, buffer depths 1
    (every step waits for the one before) and 3: seven steps with a different capture each, every record equal to the one-call-at-a-time
    reference; fetch() of the newest `depth` steps only."""
    import torch
    from amaranth_twstft_amd.wideband import WidebandSession, godual_plan
    dev = torch.device("cuda", 0)
    nchips, n, W, dec, sps_in = 5000, 10000, 3, 14, 28
    fs_in = FS * dec
    taps = frontend.lowpass_taps(fs_in, 2.1e6, 3.0e6)
    assert taps.size == 57
    codes = {"A": chips_for(13, 27, nchips), "B": chips_for(14, 43, nchips)}
    cdev = {k: torch.from_numpy(v).to(dev) for k, v in codes.items()}
    n_in = (W * n - 1) * dec + taps.size
    plan = godual_plan(("A", "B"), FS, n)

    def capture(st, other, seed):
        a = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        b = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
        _synth_dev(a, n_in, cdev[st], nchips, sps_in, [synth.SynthParams(delay_q8=(1000 + 977 * seed) * 256, fstep=synth.fstep_for_df(3.25, fs_in), phi0=9, amp=2500,
                                                                       noise_gain=synth.noise_gain_for_sigma(1500.0), seed=500 + seed, stream=0)])
        _synth_dev(b, n_in, cdev[other], nchips, sps_in, [synth.SynthParams(delay_q8=(70000 + 1201 * seed) * 256, fstep=synth.fstep_for_df(50_000.0 if st == "A" else -50_000.0, fs_in),
                                                                          phi0=7, amp=1500, noise_gain=0, seed=600 + seed, stream=1)])
        torch.cuda.synchronize()
        out = (a.to(torch.int32) + b.to(torch.int32)).clamp_(-32768, 32767).to(torch.int16).contiguous()
        torch.cuda.synchronize()
        return out

    steps = [{"A": capture("A", "B", 2 * i), "B": capture("B", "A", 2 * i + 1)} for i in range(7)]
    key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df, r.correction, r.SNRr, r.SNRi)
    ref = []
    nar = torch.empty((W * n, 2), dtype=torch.int16, device=dev)
    with Correlator(codes["A"], fs=FS, Nint=1) as ca, Correlator(codes["B"], fs=FS, Nint=1) as cb:
        by_code = {"A": ca, "B": cb}
        for cs in steps:
            out = {}
            for name, (st, code_st, band) in plan.items():
                c = by_code[code_st]
                assert c.fir_decimate_dev(cs[st].data_ptr(), n_in, taps, dec, out_i16_dev=nar.data_ptr()) == W * n
                c.synchronize()
                out[name] = [key(r) for r in _records_dev(c, nar, W, band)]
            ref.append(out)
    assert all(ref[i] != ref[i + 1] for i in range(6))
    with WidebandSession(codes, taps, dec, fs=FS, windows=W, depth=depth) as sess:
        assert sess.plan == plan and sess.n_in == n_in
        for i, cs in enumerate(steps):
            assert sess.submit({st: t.data_ptr() for st, t in cs.items()}) == i
            if depth == 1 or i >= 2:
                j = i if depth == 1 else i - 2                       # depth 3: fetched two steps behind
                got = sess.fetch(j)
                assert {k_: [key(r) for r in v] for k_, v in got.items()} == ref[j], j
        if depth == 3:
            for j in (5, 6):
                assert {k_: [key(r) for r in v] for k_, v in sess.fetch(j).items()} == ref[j], j
            with pytest.raises(ValueError):
                sess.fetch(3)
        with pytest.raises(ValueError):
            sess.fetch(7)


def _records_dev(c, nar, W, band):
    import torch
    o = torch.zeros((W, C.sizeof(L.twx_result)), dtype=torch.uint8, device=nar.device)
    b = L.twx_band(*band)
    L.check(c._lib.twx_process_windows_dev(c._h, nar.data_ptr(), W, 1, 0, C.byref(b), None, o.data_ptr()), c._h)
    c.synchronize()
    return list((L.twx_result * W).from_buffer_copy(o.cpu().numpy().tobytes()))


# --------------------------------------------------------------------------------------------------------------
# multi-rank path with the real correlator (ranks share GPU 0, records exchanged with gloo)
# --------------------------------------------------------------------------------------------------------------
def _write_capture(tmp_path, nchips, nwin, seed, bitlen=14):
    from tests.test_gpu_parity import _capture
    chips, raw = _capture(bitlen, 43, nchips, nwin, seed=seed)
    (tmp_path / "codes").mkdir()
    prn.lfsr_chips(bitlen, 43, nchips).tofile(tmp_path / "codes" / f"noiselen{nchips}_bitlen{bitlen}_taps43.bin")
    prn.lfsr_chips(bitlen, 57, nchips).tofile(tmp_path / "codes" / f"noiselen{nchips}_bitlen{bitlen}_taps57.bin")
    raw.tofile(tmp_path / "1670074501.bin")
    return chips, raw


@pytest.mark.parametrize("world,nwin", [(2, 13), (8, 600)])
def test_ranks_share_one_capture_and_match_single_rank(tmp_path, world, nwin):
    """python -m amaranth_twstft_amd.godual_ranging --gpus N: every rank correlates its contiguous block of windows of
    the SAME capture file with the real HIP correlator, one all_gather of the records, rank 0 writes .mat/TSV.
    Compared with the single-process run: same stdout, same .mat bytes.  (8 x 75 = configs[3]'s sharding.)"""
    _write_capture(tmp_path, 10000, nwin, seed=77)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "amaranth_twstft_amd.godual_ranging", "--datalocation", str(tmp_path), "--codelocation", str(tmp_path / "codes")]
    one = subprocess.run(base, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    mat = tmp_path / "1670074501.mat"
    ref_bytes = mat.read_bytes()[128:]                                 # MAT-v5 header (128 B) carries the creation time
    mat.unlink()
    many = subprocess.run(base + ["--gpus", str(world), "--backend", "gloo"], capture_output=True, text=True, env=env, timeout=900)
    assert many.returncode == 0, many.stdout[-2000:] + many.stderr[-3000:]
    assert mat.read_bytes()[128:] == ref_bytes                          # every variable, byte for byte
    rows = lambda s: [l for l in s.splitlines() if l[:1].isdigit() and "\t" in l]
    assert rows(many.stdout) == rows(one.stdout) and len(rows(one.stdout)) == nwin


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without torchrun starts two ranks itself (they share GPU 0 here: gloo) and prints n_gpus 2."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "9",
                          "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["integer_lag_exact"] and j["value"] > 0 and j["roofline"]["frac"] > 0.05
    # the gathered buffer was checked against what each rank was given (window_params(p, rank)): both ranks' blocks exact
    c = j["collective"]
    assert c["world"] == 2 and c["records"] == 18 and c["ranks_with_exact_lags"] == 2 and c["gathered_lag_exact"]
    assert c["own_block_identical"] and c["all_ranks_agree"] and c["backend"] == "gloo"


def test_bench_eight_ranks_collective_path():
    """`python bench.py --gpus 8 --backend gloo --windows 8`: the 8-rank form of the line (configs[3]'s world size), ranks sharing the
    GPUs the box has; the gathered buffer of EVERY rank is checked against what each of the eight ranks was given."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--windows", "8",
                          "--backend", "gloo", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = j["collective"]
    assert j["n_gpus"] == 8 and j["integer_lag_exact"] and c["world"] == 8 and c["records"] == 64
    assert c["ranks_with_exact_lags"] == 8 and c["gathered_lag_exact"] and c["own_block_identical"] and c["all_ranks_agree"]


@pytest.mark.parametrize("world,windows", [(2, 9), (8, 20)])
def test_bench_strong_scaling_leg_times_configs3_as_written(world, windows):
    """The N > 1 line carries `strong_workload`: ONE recording of `--windows` windows IN TOTAL sharded over the ranks (contiguous blocks,
    dist.shard_windows: 9 = 5 + 4; 20 over 8 = 3,3,3,3,2,2,2,2), one all-gather of the padded blocks per step, the exchange timed alone —
    BASELINE.json configs[3] as written (godual_ranging.m:75-102).  The weak headline is unchanged beside it.  Ranks share GPU 0: gloo."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--windows", str(windows),
                          "--backend", "gloo", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == world and j["scaling"] == "weak" and j["integer_lag_exact"] and j["collective"]["records"] == world * windows
    s = j["strong_workload"]
    base, rem = divmod(windows, world)
    assert s["scaling"] == "strong" and s["windows_total"] == windows and s["windows_per_rank"] == [base + (1 if r < rem else 0) for r in range(world)]
    assert s["ranks_with_exact_lags"] == world and s["own_block_identical"] and s["all_ranks_agree"]
    assert s["value"] > 0 and s["ms_per_step"] > 0 and s["compute_only_ms_per_step"] > 0 and s["gather_ms"] > 0
    assert s["collective"]["backend"] == "gloo" and s["collective"]["records"] == world * (base + (1 if rem else 0))
    assert 0.0 <= s["exchange_share_of_step"] < 1.0 and s["ratio_to_weak_headline"] > 0


def test_bench_one_rank_line_has_no_strong_leg():
    """N = 1: the line is the round-5 line (no `strong_workload`, `scaling` weak)."""
    j, _ = _bench_line(["--steps", "2", "--warmup", "1", "--windows", "9", "--no-cpu-baseline", "--no-roofline"])
    assert j["n_gpus"] == 1 and "strong_workload" not in j and j["scaling"] == "weak" and j["integer_lag_exact"]


def test_bench_two_ranks_over_real_rccl():
    """`python bench.py --gpus 2` with the nccl (= RCCL) backend, one GPU per rank: runs wherever the box has two GPUs
    (the driver's 8-GPU node), skipped on a one-GPU box.  The gathered records of BOTH ranks must carry their own lags."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL ranks cannot share a device")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "24",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = j["collective"]
    assert j["n_gpus"] == 2 and j["integer_lag_exact"] and c["backend"].startswith("nccl")
    assert c["records"] == 48 and c["ranks_with_exact_lags"] == 2 and c["gathered_lag_exact"] and c["all_ranks_agree"]


def _bench_line(args, env_extra=None, torchrun=0, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, **(env_extra or {}))
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={torchrun}", "--master-addr", "127.0.0.1",
            "--master-port", "29541"] if torchrun else [sys.executable]
    out = subprocess.run(head + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]), out


@pytest.mark.parametrize("how", ["driver", "self", "real_rccl_error", "probe_child", "job"])
def test_bench_asked_for_rccl_never_ends_empty(how):
    """`bench.py --gpus 2` with the default backend (nccl = RCCL) where RCCL cannot come up must still print a valid line, exit 0,
    with the exchange on gloo and the reason in `collective.backend`:
    driver  — started the way the round-end driver starts it (torch.distributed.run ... bench.py --gpus 2); two ranks, one GPU here:
              rank 0's probe refuses before anything touches the device;
    self    — the self-launching form;
    real_rccl_error — the probe's child job really runs, its ranks sharing the GPU: RCCL itself refuses (a genuine RCCL error text);
    probe_child     — one rank of the probe job dies (wherever >= 2 GPUs exist this is the injected form of the same);
    job     — a rank of the JOB dies while RCCL is asked for: the launcher, which never touched a GPU, starts ONE fresh job on gloo."""
    import torch
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "9", "--no-cpu-baseline", "--no-roofline"]
    two = torch.cuda.device_count() >= 2
    env, torchrun = {}, 0
    if how == "driver":
        torchrun = 2
        if two:
            env = {"TWX_INJECT_RCCL_FAIL": "probe"}
    elif how == "self":
        if two:
            env = {"TWX_INJECT_RCCL_FAIL": "probe"}
    elif how == "real_rccl_error":
        if two:
            pytest.skip("two GPUs: RCCL works, nothing to refuse (the injected forms cover the branch)")
        env = {"TWX_RCCL_PROBE_SHARE": "1", "TWX_RCCL_PROBE_TIMEOUT_S": "150"}
    elif how == "probe_child":
        env = {"TWX_RCCL_PROBE_SHARE": "1", "TWX_INJECT_RCCL_FAIL": "probe_child:1", "TWX_RCCL_PROBE_TIMEOUT_S": "150"}
    else:
        env = {"TWX_INJECT_RCCL_FAIL": "exit:1"}
    j, out = _bench_line(args, env, torchrun, timeout=1500)
    c = j["collective"]
    assert j["n_gpus"] == 2 and j["integer_lag_exact"] and j["value"] > 0
    assert c["backend"].startswith("gloo (fallback: "), c
    assert c["requested"] == ("gloo" if how == "job" else "nccl")
    assert c["world"] == 2 and c["records"] == 18 and c["ranks_with_exact_lags"] == 2 and c["gathered_lag_exact"] and c["all_ranks_agree"]
    assert len(c["numa"]) == 2 and all(x["ok"] for x in c["numa"])
    assert "cpu_baseline" in j["omitted_at_n_gt_1"]
    if how == "job":
        assert "restarted by the launcher" in c["backend"] and "starting it once more" in out.stderr
    elif how in ("real_rccl_error", "probe_child"):
        assert "RCCL probe job failed" in c["backend"] and c["rccl_probe"]["ok"] is False and len(c["rccl_probe"]["tried"]) == 2   # both environments tried
    else:
        assert "RCCL probe job failed" in c["backend"]


@pytest.mark.parametrize("step", ["init:0", "rehearsal:0"])
def test_bench_rccl_bring_up_failure_in_the_rank_falls_back(step):
    """The in-rank stages after a good probe: the RCCL sub-group's first all_reduce, then the rehearsal of the real gather — a failure
    in either (injected; world of one, --force-dist) is agreed on over the control plane and the exchange stays on gloo."""
    j, _ = _bench_line(["--gpus", "1", "--steps", "2", "--warmup", "1", "--windows", "9", "--force-dist", "--no-cpu-baseline", "--no-roofline"],
                       {"TWX_INJECT_RCCL_FAIL": step, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534"})
    c = j["collective"]
    assert c["backend"].startswith("gloo (fallback: RCCL bring-up failed: rank 0: RuntimeError: injected failure"), c
    assert j["integer_lag_exact"] and c["gathered_lag_exact"] and c["own_block_identical"] and c["all_ranks_agree"]


def test_matrix_core_fir_never_runs_beside_a_correlation():
    """TWX_OPT_FIR_MFMA: k_fir_mfma makes packed-fp32 results of co-resident waves go wrong (profiles/r05_fir_mfma.txt: 6-12 wrong spectrum
    rows of k_rowd per call, 7-8 wrong records of 12).  The library orders every such launch behind all other work it has enqueued on the
    device and all later work behind it (csrc/twx_internal.h), so the SAME harness — a correlation enqueued right behind a matrix-core FIR of
    another context, nothing synchronised in between — now returns the records of the correlation run alone, 24 of 24, from two host threads
    as well; the session forces the vector form on its contexts whatever the environment says."""
    import threading
    import torch
    from amaranth_twstft_amd import frontend, prn
    lib = L.load()
    dev = torch.device("cuda", 0)
    Nw, dec = 5_000_000, 14
    taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
    n_in = (Nw - 1) * dec + taps.size
    chips = prn.lfsr_chips(22, 3, 2_500_000)
    g = torch.Generator(device=dev); g.manual_seed(1)
    cap = (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16)
    win = [(torch.randn((Nw, 2), device=dev, generator=g) * 4000).to(torch.int16) for _ in range(2)]
    out16 = torch.zeros((Nw, 2), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    band = L.twx_band(*band_godual(FS, Nw))
    key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df, r.SNRr)
    RB = C.sizeof(L.twx_result)
    with Correlator(chips, fs=FS, Nint=1) as c, Correlator(chips, fs=FS, Nint=1) as b1:
        L.check(lib.twx_set_option(b1._h, L.TWX_OPT_FIR_MFMA, 1), b1._h)
        chain = lambda i, r: L.check(lib.twx_process_windows_dev(c._h, win[i % 2].data_ptr(), 1, 1, 0, C.byref(band), None, r.data_ptr()), c._h)
        alone = []
        for i in range(2):
            r = torch.zeros(RB, dtype=torch.uint8, device=dev); chain(i, r); c.synchronize()
            alone.append(key(L.twx_result.from_buffer_copy(r.cpu().numpy().tobytes())))
        res = torch.zeros((24, RB), dtype=torch.uint8, device=dev)
        for i in range(12):                                     # one host thread: FIR of context b1, then the chain of context c, back to back
            b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out16.data_ptr())
            chain(i, res[i])
        stop = threading.Event()

        def fir_loop():                                         # a second host thread keeps matrix-core FIRs coming
            torch.cuda.set_device(0)
            while not stop.is_set():
                b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out16.data_ptr())
                b1.synchronize()
        th = threading.Thread(target=fir_loop); th.start()
        try:
            for i in range(12, 24):
                chain(i, res[i]); c.synchronize()
        finally:
            stop.set(); th.join()
        c.synchronize(); b1.synchronize(); torch.cuda.synchronize()
        host = res.cpu().numpy()
        bad = [i for i in range(24) if key(L.twx_result.from_buffer_copy(host[i].tobytes())) != alone[i % 2]]
        assert not bad, "records that differ from the correlation run alone: %r" % bad
        # the FIR's own output is the vector form's to one count
        ref16 = out16.clone()
        L.check(lib.twx_set_option(b1._h, L.TWX_OPT_FIR_MFMA, 0), b1._h)
        b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out16.data_ptr()); b1.synchronize()
        assert int((out16.to(torch.int32) - ref16.to(torch.int32)).abs().max().item()) <= 1


def test_bench_wideband_and_fp64_legs():
    """`bench.py --wideband-only`: the configs[4] objects of the driver's line — 70 Msps -> FIR -> four correlations in flight together
    (fp32) with the FIR's fp32-vector roofline, and the same four in fp64 with the roofline of their dominant kernel; lags as expected,
    fp32 = fp64 lag for lag, |peak| within 1e-6."""
    j, _ = _bench_line(["--wideband-only", "--wideband-seconds", "1"])
    w, f = j["wideband_workload"], j["f64_workload"]
    assert w["expected_lags_within_one_sample"] and w["input_Msamples_per_s"] > 1000 and w["correlated_Msamples_per_s"] > 1000
    assert w["fir"]["roofline"]["bound"] == "fp32 vector" and 0.05 < w["fir"]["roofline"]["frac"] < 1.0 and w["fir"]["avg_ms"] > 0
    mc = w["fir"]["matrix_core_form"]                                  # the opt-in form, measured alone: the same outputs to one count
    assert mc["kernel"].startswith("k_fir_mfma") and mc["avg_ms"] > 0 and mc["max_abs_difference_from_the_vector_form_int16"] <= 1
    assert w["ms_per_step"] > 0 and w["ms_per_step_one_at_a_time"] > 0
    assert f["dtype"] == "f64" and f["integer_lags_equal_fp32"] and f["within_tolerance"] and f["fp32_vs_fp64_peak_rel"] <= 1e-6
    r = f["roofline"]
    assert r["bound"] == "hbm" and r["kernel"].startswith("k_row_mid") and 0.05 < r["frac"] < 1.0 and r["algorithmic_bytes_per_launch"] == 64 * 5_000_000 + 16 * 5_000_000


def test_bench_measures_the_pmc_traffic_itself():
    """roofline.traffic of the driver-style line is measured by the invocation (two rocprofv3 --pmc child passes before the first
    GPU call), not read from a committed file; it equals the dominant kernel's algorithmic bytes to 2 % (no wasted re-reads)."""
    import shutil
    if not shutil.which("rocprofv3"):
        pytest.skip("rocprofv3 not on PATH")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in list(env):
        if k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")):
            env.pop(k)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--windows", "24",
                          "--no-cpu-baseline", "--no-caf"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = j["roofline"]
    if "traffic_live_error" in r:                                    # the profiler could not run here: the line falls back to the committed file
        pytest.skip("rocprofv3 --pmc child pass unavailable: " + r["traffic_live_error"][:200])
    assert r["traffic_source"].startswith("measured by this invocation"), r
    assert abs(r["traffic"] / r["algorithmic_bytes_per_launch"] - 1.0) < 0.02, r


def test_bench_rccl_calls_with_a_world_of_one():
    """The N > 1 path of bench.py talks to RCCL (backend "nccl"): process group bound to the device, all_gather_into_tensor of
    the uint8 result records, all_reduce(MAX) of the step time, barrier.  Two ranks cannot share one GPU under RCCL, so the
    calls themselves run here with one rank (--force-dist); the two-rank logic runs above on gloo."""
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--windows", "9",
                          "--force-dist", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["integer_lag_exact"] and j["value"] > 0
    c = j["collective"]
    assert c["backend"].startswith("nccl") and c["world"] == 1 and c["gathered_lag_exact"] and c["own_block_identical"]
    assert j["startup_s"]["to_first_step"] > 0


# --------------------------------------------------------------------------------------------------------------
# single-slot ingest pipeline (TWX_FLAG_PROFILE forces one slot; TWX_STREAMS=1)
# --------------------------------------------------------------------------------------------------------------
def test_single_slot_pipeline_matches_three_slots(tmp_path, monkeypatch):
    from tests.test_gpu_parity import _capture
    chips, raw = _capture(15, 3, 25000, 11, seed=93)
    n = 50000
    path = tmp_path / "1670074999.bin"
    np.concatenate([raw, raw[:777]]).tofile(path)
    band = band_godual(FS, n)
    with Correlator(chips, fs=FS, Nint=1, max_batch=2) as cor:
        ref_mem = cor.process(raw, n_channels=2, channel=0, band=band)
        ref_fil = cor.process_file(str(path), n_channels=2, channel=-1, band=band)
    with Correlator(chips, fs=FS, Nint=1, max_batch=2, profile=True) as cor:
        got_mem = cor.process(raw, n_channels=2, channel=0, band=band)
        got_fil = cor.process_file(str(path), n_channels=2, channel=-1, band=band)
        assert cor.profile()["k_row_mid"]["launches"] > 0
    monkeypatch.setenv("TWX_STREAMS", "1")
    with Correlator(chips, fs=FS, Nint=1, max_batch=2) as cor:
        got2_mem = cor.process(raw, n_channels=2, channel=0, band=band)
        got2_fil = cor.process_file(str(path), n_channels=2, channel=-1, band=band)
        one = cor.process(raw[:n], n_channels=2, channel=1, df=0.0)            # a single window, one chunk
    assert len(ref_mem) == 11 and len(one) == 1
    for got in (got_mem, got2_mem):
        assert [(g.indice, g.xval, g.df, g.SNRr) for g in got] == [(g.indice, g.xval, g.df, g.SNRr) for g in ref_mem]
    for got in (got_fil, got2_fil):
        for c in (0, 1):
            assert [(g.indice, g.xval, g.df) for g in got[c]] == [(g.indice, g.xval, g.df) for g in ref_fil[c]]


# --------------------------------------------------------------------------------------------------------------
# the reference's own signatures: processing(d,k) / processing(d,df) on the complex double column d
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nchips,bitlen", [(10000, 14), (100000, 17)])
def test_processing_complex_signature(nchips, bitlen):
    """twx_process_complex = processing(d,k) of godual_ranging.m:12 as the script calls it: d complex double, mean
    removed by the caller (:80).  Same results as the oracle on the same d, and as the raw-int16 entry point."""
    from tests.test_gpu_parity import _capture, _check
    chips, raw = _capture(bitlen, 43 if bitlen == 14 else 9, nchips, 3, seed=55)
    n = 2 * nchips
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    freq = orc.freq_axis(FS, n)
    k = orc.band_godual(freq)
    temps = np.arange(n) / FS
    ds = []
    for w in range(3):
        d = orc.deinterleave(raw[w * n:(w + 1) * n], 2, 0)
        ds.append(d - d.mean())
    dcat = np.concatenate(ds)
    with Correlator(chips, fs=FS, Nint=1) as cor:
        got = cor.processing_complex(dcat, k=k)                       # three windows in one call (vector outputs)
        got_df = cor.processing_complex(ds[1], df=1780.75)
        raw_path = cor.process(raw, n_channels=2, channel=0, band=(int(k[0]), int(k[-1])))
    for w in range(3):
        o = orc.processing(ds[w], k, freq, temps, fcode, code, Nint=1, fs=FS)
        _check(got[w], o)
        assert got[w].indice == raw_path[w].indice and abs(abs(got[w].xval) - abs(raw_path[w].xval)) <= MAG_TOL * abs(got[w].xval)
    o = orc.processing(ds[1], None, None, temps, fcode, code, Nint=1, fs=FS, df=1780.75)
    _check(got_df[0], o)
    # processing(d,df) in the claudio convention (fcode.*conj(ffty), claudio_aligned_code_ranging_separate.m:49-102), Octave var
    fc = orc.make_fcode(code, "claudio")
    with Correlator(chips, fs=FS, Nint=1, convention="claudio", var_ddof=1) as cor:
        g = cor.processing_complex(ds[2], df=1780.75)[0]
    oc = orc.processing_claudio(ds[2], 1780.75, temps, fc, code, Nint=1, ddof=1)
    _check(g, oc)
    # a caller that did NOT remove the mean gets the result of the un-centred d (the library does not touch it)
    dm = orc.deinterleave(raw[:n], 2, 0)
    with Correlator(chips, fs=FS, Nint=1, precision="f64") as c64:
        g64 = c64.processing_complex(dm, df=1780.75)[0]
    om = orc.processing(dm, None, None, temps, fcode, code, Nint=1, fs=FS, df=1780.75)
    assert g64.indice == om["indice"] and abs(g64.puissance - om["puissance"]) <= 1e-9 * om["puissance"]
    assert abs(g64.xval - om["xval"]) <= 1e-9 * abs(om["xval"])


def _read_mex_outputs(path):
    b = open(path, "rb").read()
    cnt = int(np.frombuffer(b, np.int32, 1, 0)[0])
    off, outs = 4, []
    for _ in range(cnt):
        m, n, cplx = (int(v) for v in np.frombuffer(b, np.int32, 3, off))
        off += 12
        re = np.frombuffer(b, np.float64, m * n, off); off += 8 * m * n
        if cplx:
            im = np.frombuffer(b, np.float64, m * n, off); off += 8 * m * n
            re = re + 1j * im
        outs.append(re.reshape(n, m).T)              # column-major m x n
    return outs


def test_mex_gateway_runs_and_matches_the_ctypes_path(tmp_path):
    """mexFunction() itself, executed on the GPU box on top of the functional fake mex.h: the reference's two signatures
    (processing(d,k) in the godual output order, processing(d,df) in the claudio order with complex d) and the raw-int16
    form, against the same calls through ctypes."""
    import subprocess
    from tests.test_abi_and_host import build_mex_harness
    from tests.test_gpu_parity import _capture
    exe = build_mex_harness(ROOT, tmp_path)
    nchips, n, nwin = 10000, 20000, 4
    chips, raw = _capture(14, 43, nchips, nwin, seed=63)
    raw.tofile(tmp_path / "cap.bin")
    chips.tofile(tmp_path / "chips.bin")
    band = band_godual(FS, n)
    common = [str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"), "2"]

    def run(mode, chan, kdf, *conv):
        r = subprocess.run([str(exe), mode, *common, str(chan), *kdf, "5e6", "1", *conv], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return _read_mex_outputs(tmp_path / "out.bin")

    ds = []
    for w in range(nwin):
        d = orc.deinterleave(raw[w * n:(w + 1) * n], 2, 0)
        ds.append(d - d.mean())
    with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as cor, Correlator(chips, fs=FS, Nint=1, var_ddof=1, convention="claudio") as cc:
        ref_k = cor.processing_complex(np.concatenate(ds), k=band)
        ref_cl = cc.processing_complex(np.concatenate(ds), df=1780.75)
        ref_raw = cor.process(raw, n_channels=2, channel=ALL_CHANNELS, band=band)
    # (A) processing(d,k): indice correction SNRr SNRi df puissance puissancecode puissancenoise xval, 1 x nwin each
    o = run("complex", 1, [str(band[0] + 1), str(band[1] + 1)])
    assert len(o) == 9 and all(x.shape == (1, nwin) for x in o)
    for w, r in enumerate(ref_k):
        got = [x[0, w] for x in o]
        assert got[0] == r.indice + 1                                    # Octave 1-based
        assert (got[1], got[2], got[3], got[4], got[5], got[6], got[7]) == (r.correction, r.SNRr, r.SNRi, r.df, r.puissance, r.puissancecode, r.puissancenoise)
        assert got[8] == r.xval
    # (A') processing(d,df), claudio: xval indice correction SNRr SNRi puissance puissancecode puissancenoise [df]
    o = run("complex", 1, ["df", "1780.75"], "claudio")
    for w, r in enumerate(ref_cl):
        got = [x[0, w] for x in o]
        assert got[0] == r.xval and got[1] == r.indice + 1 and got[2] == r.correction and got[5] == r.puissance and got[8] == 1780.75
    oc = orc.processing_claudio(ds[0], 1780.75, np.arange(n) / FS, orc.make_fcode(orc.make_code(chips, 2), "claudio"), orc.make_code(chips, 2), Nint=1, ddof=1)
    assert o[1][0, 0] == oc["indice"] + 1 and abs(abs(o[0][0, 0]) - abs(oc["xval"])) <= MAG_TOL * abs(oc["xval"])
    # (B) raw int16, every channel from one upload: nchan x nwin
    o = run("raw", 0, [str(band[0] + 1), str(band[1] + 1)])
    assert all(x.shape == (2, nwin) for x in o)
    for c in (0, 1):
        for w, r in enumerate(ref_raw[c]):
            assert o[0][c, w] == r.indice + 1 and o[8][c, w] == r.xval and o[4][c, w] == r.df


# --------------------------------------------------------------------------------------------------------------
# window lengths outside the built-in plan list (plan plug-ins, amaranth_twstft_amd/plans.py)
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bitlen,taps,nchips,Nint", [(13, 27, 2500, 0), (13, 27, 2500, 1), (15, 3, 12500, 1), (16, 45, 32768, 1), (13, 27, 7000, 1)])
def test_plan_plugins_generic_lengths(bitlen, taps, nchips, Nint):
    """BASELINE.json configs[0] (1-ms window: N = 5000, first 2500 chips of LFSR(13, 27); SURVEY §8d C1: delay 1234,
    A = 300, sigma = 600, code-phase-only, Nint 0 and 1), an odd code length (N = 25000) and a power-of-two window
    (N = 65536) and one with a factor 7 (N = 14000: radix-7 butterflies): none of them is in the built-in plan list,
    all run through plug-ins of the same kernels."""
    from tests.test_gpu_parity import _check
    from amaranth_twstft_amd import plans
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    assert plans.choose(n) is not None
    p = synth.SynthParams(delay_q8=1234 * 256, fstep=synth.fstep_for_df(0.0 if nchips == 2500 else 977.5, FS), phi0=0, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(600.0), seed=1)
    raw = synth.synth_channel(3 * n, chips, 2, p)
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    freq = orc.freq_axis(FS, n)
    k = orc.band_numpy(freq, 0.0, 8000.0)
    temps = np.arange(n) / FS
    with Correlator(chips, fs=FS, Nint=Nint) as cor:
        assert cor.info.n == n and cor.info.n1 * cor.info.n2 == n
        got = cor.process(raw, n_channels=1, channel=0, df=0.0) if nchips == 2500 else \
            cor.process(raw, n_channels=1, channel=0, band=(int(k[0]), int(k[-1])))
        spec = cor.code_spectrum()
        x = (np.arange(n) % 7 - 3) + 1j * (np.arange(n) % 5 - 2)
        f = cor.fft(x)
    assert np.abs(spec - fcode).max() <= 2e-7 * np.abs(fcode).max()                 # computed in fp64 through the plug-in plans, held in fp32
    ref = np.fft.fft(x)
    assert np.abs(f - ref).max() <= 3e-6 * np.abs(ref).max()
    for w in range(3):
        d = orc.deinterleave(raw[w * n:(w + 1) * n], 1, 0)
        d = d - d.mean()
        o = orc.processing(d, k, freq, temps, fcode, code, Nint=Nint, fs=FS, df=0.0 if nchips == 2500 else None)
        _check(got[w], o)
        assert got[w].indice == (2 * Nint + 1) * 1234


def test_contexts_survive_later_plan_registrations():
    """A live context keeps pointers to its column/row plans; plug-ins loaded afterwards (another code length) append to
    the same registry.  The first context must keep working (the registry never moves its entries)."""
    from amaranth_twstft_amd import plans
    chips_a = chips_for(14, 43, 10000)                    # N = 20 000: built-in plans
    n_a = 20000
    p = synth.SynthParams(delay_q8=4321 * 256, fstep=synth.fstep_for_df(410.0, FS), phi0=3, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=9)
    raw = synth.synth_channel(2 * n_a, chips_a, 2, p)
    band = band_godual(FS, n_a)
    with Correlator(chips_a, fs=FS, Nint=1) as a:
        before = a.process(raw, n_channels=1, channel=0, band=band)
        others = []
        for nchips in (2500, 12500, 7000, 3000, 6000, 9000):      # lengths outside the built-in list (plug-ins pre-built by build())
            n = 2 * nchips
            assert plans.choose(n) is not None
            others.append(Correlator(chips_for(15, 3, nchips), fs=FS, Nint=1))
        after = a.process(raw, n_channels=1, channel=0, band=band)
        for o in others:
            o.close()
    for x, y in zip(before, after):
        assert x.indice == y.indice == 3 * 4321 and x.xval == y.xval and x.df == y.df


def test_native_70msps_window_without_decimation():
    """BASELINE.json configs[4] read the other way: the 70 Msps capture correlated at its NATIVE rate, one window of
    N = 7e7 samples (28 samples per chip, 2^7 5^7 7: a 7000 x 10000 plan with radix-14 columns, fp32 only: the complex-double
    forms do not fit the LDS).  Gates: lag = 3 x the generator's delay, the same lag as the FIR + decimate-by-14 route sees
    (configs[4] test above), and - when the host has the memory for a 2.1e8-point numpy ifft - bit-exact lag and 1e-6
    |peak| against the oracle on the same samples."""
    import psutil
    import torch
    dev = torch.device("cuda", 0)
    fs, sps, n = 70e6, 28, 70_000_000
    chips = chips_for(22, 57, NCHIPS)
    d0 = 18_364_717
    p = synth.SynthParams(delay_q8=d0 * 256, fstep=synth.fstep_for_df(3.25, fs), phi0=99, amp=2500,
                          noise_gain=synth.noise_gain_for_sigma(2500.0), seed=401)
    wide = torch.empty((n, 2), dtype=torch.int16, device=dev)
    _synth_dev(wide, n, torch.from_numpy(chips).to(dev), NCHIPS, sps, [p])
    torch.cuda.synchronize()
    band = band_godual(fs, n)
    with Correlator(chips, fs=fs, sps=sps, Nint=1) as cor:
        assert cor.info.n == n and cor.info.n1 * cor.info.n2 == n
        g = cor.process_dev(wide.data_ptr(), 1, band=band)[0]
        x = (np.arange(n) % 7 - 3) + 1j * (np.arange(n) % 5 - 2)
        f = cor.fft(x)
    assert g.indice == 3 * d0
    assert abs(g.df - 3.25) <= 0.5 + 1e-9                       # the squared-spectrum arg-max resolves fs/(2N) = 0.5 Hz
    assert abs(g.correction) < 0.5
    idx = np.concatenate([np.arange(0, 4096), np.random.default_rng(5).integers(0, n, 4096)])
    # the transform itself against a direct evaluation of 8192 output bins of a sparse-support input would need the full
    # FFT; use numpy's (fp64) on the same input
    ref = np.fft.fft(x)
    assert np.abs(f[idx] - ref[idx]).max() <= 3e-6 * np.abs(ref).max()
    del ref, f, x
    if psutil.virtual_memory().available < 56 * 2 ** 30:
        pytest.skip("lag and transform checked; the fp64 oracle at N = 7e7 needs ~40 GiB of host memory")
    raw = wide.cpu().numpy()
    del wide
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    code = orc.make_code(chips, sps)
    k = np.arange(band[0], band[1] + 1)
    o = orc.processing(d, k, orc.freq_axis(fs, n), np.arange(n) / fs, orc.make_fcode(code), code, Nint=1, fs=fs)
    assert g.indice == o["indice"]
    assert abs(g.df - o["df"]) <= 1e-9
    assert abs(abs(g.xval) - abs(o["xval"])) <= MAG_TOL * abs(o["xval"])
    assert abs(g.correction - o["correction"]) <= 2e-4


# --------------------------------------------------------------------------------------------------------------
# a11: the acquisition stage of experiments/231001_DLL_PLL/rxcomplex.cpp with the program's own arithmetic, sdr.param sizes
# --------------------------------------------------------------------------------------------------------------
def test_rxcomplex_acquisition_pipeline_at_sdr_param_sizes():
    """1-s buffer of 2-channel int16 at 5 Msps -> short2double x2 interpolation (10 Msps) -> per trial carrier
    downconv_acq / FFT(2^20) / cross_spectrum / IFFT / izamax -> coarse sweep + step halving (rxcomplex.cpp:469-573),
    code = 100 kchip LFSR(17, taps 9) = experiments/231001_DLL_PLL/0.bin at 2.5 Mchip/s (sdr.param: nobs = 400 000, nfft = 2^20).
    Device against the line-by-line oracle restatement (unpinned: the C++ program cannot be built here)."""
    import torch
    from amaranth_twstft_amd import acquisition as acq
    dev = torch.device("cuda", 0)
    n_in, fs, rc, clen = 5_000_000, 10e6, 2.5e6, 100_000
    nobs = int(fs) // 25
    chips = {"A": chips_for(17, 9, clen), "B": chips_for(17, 15, clen)}
    d0, fc_true = 123_457, 1307.25                                   # delay in 5-Msps samples, carrier offset (Hz)
    chans = [synth.SynthParams(delay_q8=d0 * 256, fstep=synth.fstep_for_df(fc_true, 5e6), phi0=11, amp=400,
                               noise_gain=synth.noise_gain_for_sigma(900.0), seed=31, stream=0),
             synth.SynthParams(delay_q8=77_001 * 256, fstep=synth.fstep_for_df(-260.0, 5e6), phi0=5, amp=500,
                               noise_gain=synth.noise_gain_for_sigma(700.0), seed=31, stream=1)]
    # two different codes on the two physical channels: generate separately (the generator takes one code per call)
    iq = torch.empty((n_in, 4), dtype=torch.int16, device=dev)
    tmp = torch.empty((n_in, 2), dtype=torch.int16, device=dev)
    for c, key in enumerate(("A", "B")):
        _synth_dev(tmp, n_in, torch.from_numpy(chips[key]).to(dev), clen, 2, [chans[c]])
        torch.cuda.synchronize()
        iq[:, 2 * c:2 * c + 2] = tmp
    raw = iq.cpu().numpy()
    smp_dev = torch.empty((2 * n_in, 2), dtype=torch.float32, device=dev)
    interp = acq.Interpolator(n_in)
    interp(iq.data_ptr(), smp_dev.data_ptr(), n_channels=2, channel=0)
    interp.cor.synchronize()
    # ---- oracle: interpolation
    oA, _ = orc.rx_short2double(raw.reshape(-1), 2 * n_in)
    got = smp_dev.cpu().numpy()
    got = got[:, 0].astype(np.float64) + 1j * got[:, 1]
    scale = np.abs(oA).max()
    assert np.abs(got - oA).max() <= 2e-6 * scale
    assert np.abs(oA[0::2] - 0).max() > 0 and abs(np.abs(oA[2 * 1000]) - 0) >= 0     # (streams are complex, both phases filled)
    # ---- replica and per-bin body
    code_pm1 = 1 - 2 * chips["A"].astype(np.int64)                                     # SDRcode: host_code = 1-2*byte (:879)
    a = acq.Acquisition(code_pm1, rc, fs, nobs)
    assert a.nfft == 1 << 20 and a.nobs == 400_000
    wav_f, psbb, _ = orc.rx_replica(code_pm1, nobs, a.nfft, rc, fs, clen, rc, -rc)
    assert np.abs(a.wav_acq_f - wav_f).max() <= 1e-9 * np.abs(wav_f).max()
    assert abs(a.psbb - psbb) <= 1e-9 * psbb
    idx = 3 * nobs                                                                       # code-aligned offset (:529 draws it at random)
    trial = [fc_true - 700.0, 1024.0, 1280.0, 1307.0, 1308.0, 1536.0, 186.0]
    pk, pki = a.bins(smp_dev.data_ptr(), idx, trial)
    for f, p, i in zip(trial, pk, pki):
        po, io = orc.rx_acq_bin(oA, idx, f, wav_f, a.nfft, fs, rc, -rc)
        assert i == io, f                                                               # cblas_izamax index, bit-exact
        assert abs(p - po) <= 3e-6 * po, f
    # ---- the sweep, reduced range for the oracle (17 + 24 bins), then sdr.param's own range on the device alone
    fc, pkb, pt = a.acquire(smp_dev.data_ptr(), idx, fc_init=1186.0, frange=2048.0, fstep=256.0)
    fo, pko, pto = orc.rx_acquire(oA, idx, wav_f, nobs, a.nfft, fs, 1186.0, 2048.0, 256.0, rc, -rc)
    assert (fc, pt) == (fo, pto) and abs(pkb - pko) <= 3e-6 * pko
    assert abs(fc - fc_true) <= 1.0 and abs(pt - (2 * d0) % nobs) <= 1           # the generator's delay, in samples of the x2 stream
    fc2, pk2, pt2 = a.acquire(smp_dev.data_ptr(), idx, fc_init=186.0, frange=65536.0, fstep=256.0)    # sdr.param: 64096 -> 65536, 256
    assert abs(fc2 - fc_true) <= 1.0 and pt2 == pt
    px = float((smp_dev.double() ** 2).sum().item()) / fs                                # received power :481-489
    assert abs(px - orc.rx_power(oA, fs)) <= 1e-5 * px
    p_sig, locked = a.gate(pk2, px, 10 ** (-18 / 10))                                    # least_required_SNR -18 dB
    assert locked and orc.rx_gate(pko, psbb, orc.rx_power(oA, fs), 10 ** (-18 / 10))[1]
    # the one-call sweep (bookkeeping between rounds on the device, one synchronisation) against the host-driven rounds
    assert a.acquire_host_loop(smp_dev.data_ptr(), idx, 1186.0, 2048.0, 256.0) == (fc, pkb, pt)
    assert a.acquire_host_loop(smp_dev.data_ptr(), idx, 186.0, 65536.0, 256.0) == (fc2, pk2, pt2) and a.n_trials == 513 + 3 * 8
    a.close()
    # ---- dec_a = 2, the B210 branch (rxcomplex.cpp:228-230,420,540-543,575): every second sample of the stream, nfft = 2^19,
    # replica decimated by memcpy_acq, carrier normalised by fs/dec_a, pt modulo nobs/dec_a
    a2 = acq.Acquisition(code_pm1, rc, fs, nobs, dec_a=2)
    assert a2.nfft == 1 << 19
    wav2, psbb2, _ = orc.rx_replica(code_pm1, nobs, a2.nfft, rc, fs, clen, rc, -rc, dec_a=2)
    assert np.abs(a2.wav_acq_f - wav2).max() <= 1e-9 * np.abs(wav2).max() and abs(a2.psbb - psbb2) <= 1e-9 * psbb2
    pk, pki = a2.bins(smp_dev.data_ptr(), idx, trial[:5])
    for f, p, i in zip(trial[:5], pk, pki):
        po, io = orc.rx_acq_bin(oA, idx, f, wav2, a2.nfft, fs, rc, -rc, dec_a=2)
        assert i == io and abs(p - po) <= 3e-6 * po, f
    fcd, pkd, ptd = a2.acquire(smp_dev.data_ptr(), idx, fc_init=1186.0, frange=1024.0, fstep=256.0)
    fod, pod, ptod = orc.rx_acquire(oA, idx, wav2, nobs, a2.nfft, fs, 1186.0, 1024.0, 256.0, rc, -rc, dec_a=2)
    assert (fcd, ptd) == (fod, ptod) and abs(pkd - pod) <= 3e-6 * pod
    assert abs(fcd - fc_true) <= 1.0 and abs(ptd * 2 - (2 * d0) % nobs) <= 2          # pt = pt*dec_a once locked (:575)
    a2.close(); interp.close()
    # a context whose batch cannot hold the three trial carriers of a refinement round refuses the sweep (it used to drop fc + step silently)
    a1 = acq.Acquisition(code_pm1, rc, fs, nobs, max_batch=2)
    with pytest.raises(L.TwxError, match="max_batch must be at least 3"):
        a1.acquire(smp_dev.data_ptr(), idx, 1186.0, 2048.0, 256.0)
    a3 = acq.Acquisition(code_pm1, rc, fs, nobs, max_batch=3)
    assert a3.acquire(smp_dev.data_ptr(), idx, 1186.0, 2048.0, 256.0) == (fc, pkb, pt)
    a1.close(); a3.close()


def test_track_epoch_on_the_device_vs_oracle_epoch():
    """twx_track_epoch_dev: ONE call from a device-resident capture to the epoch's (fc, gd, pt) update — down-conversion at the
    channel's carrier and code phase, +-28 lags x 24 code periods of 400 000 samples (sdr.param sizes), power / phase,
    high-resolution correlator, 3-sigma filter, phase unwrap, the two weighted fits (rxcomplex.cpp:593-745) — against
    oracle.rx_track_epoch on the same samples.  Three epochs in a row: the state of one feeds the next."""
    import torch
    from amaranth_twstft_amd import tracking
    dev = torch.device("cuda", 0)
    fs, rc, clen, nobs, bps, nlag = 10e6, 2.5e6, 100_000, 400_000, 25, 28
    code = 1 - 2 * chips_for(17, 9, clen).astype(np.int64)
    wav = tracking.prn_sampling(nobs, code, rc, fs)
    n = nobs * (bps + 1) + 4096
    rng = np.random.default_rng(5)
    delay, f_true = 1237, 1000.0 + 3.3
    x = np.roll(np.tile(wav.astype(np.float64), bps + 2)[:n], delay) * np.exp(2j * np.pi * (f_true / fs * np.arange(n) + 0.05))
    x = x * np.repeat(rng.integers(0, 2, bps + 2) * 2 - 1, nobs)[:n]                    # data bits: BPSK flips per code period
    raw = np.empty((n, 2), dtype=np.int16)
    raw[:, 0] = np.round(900 * x.real + rng.normal(0, 1500, n)); raw[:, 1] = np.round(900 * x.imag + rng.normal(0, 1500, n))
    xq = raw[:, 0].astype(np.float64) + 1j * raw[:, 1].astype(np.float64)
    iq = torch.from_numpy(raw).to(dev)
    rep = torch.from_numpy(wav).to(dev)
    st = dict(fc=1000.0, pt=delay - 3, last_phi=0.0, psbb=1.0, duration=nobs / fs, fs=fs)
    so = dict(st)
    with Correlator(lfsr=(14, 43, 10000), fs=5e6) as cor:
        for epoch in range(3):
            got = tracking.track_epoch_dev(cor, iq.data_ptr(), n, rep.data_ptr(), nobs, bps, nlag, st, scale=1.4142135624)
            want = orc.rx_track_epoch(xq, wav.astype(np.complex128), so, nobs, bps, nlag, fs)
            assert got is not None and want is not None and got["cnt"] == want["cnt"] >= bps - 6      # the 3-sigma filter on the IQR drops a few
            assert (st["pt"], st["fc"]) == (so["pt"], so["fc"]) and st["pt"] == delay
            assert abs(got["freq"] - want["freq"]) <= 2e-4 and abs(got["phi"] - want["phi"]) <= 1e-4
            assert abs(got["gd"] - want["gd"]) <= 0.02 and abs(got["dg"] - want["dg"]) <= 0.05 and abs(got["sdgd"] - want["sdgd"]) <= 0.05
            assert abs(got["pk"] - want["pk"]) <= 3e-6 * want["pk"]
            assert abs(got["freq"] - f_true) < 0.3


def test_aux_kernels_device_resident_forms_match_host_forms():
    """twx_sliding_dot_dev / twx_fir_decimate_dev (context stream, context-owned work buffers) against the host-pointer
    entry points on the same data, bit for bit; odd sizes exercise the ragged last workgroup of both kernels."""
    import torch
    from amaranth_twstft_amd import tracking
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    nobs, ncodes, nlag = 20000 + 37, 3, 28
    raw = np.clip(rng.normal(0, 2500, (nobs * ncodes + 5, 2)), -32768, 32767).astype(np.int16)
    rep = rng.choice([-1.0, 1.0], nobs).astype(np.float32)
    host = tracking.sliding_dot(raw, rep, nobs, ncodes, nlag, pt=3, ff=2.5e-5, phi=0.125, scale=1 / 32768)
    x = torch.from_numpy(raw).to(dev)
    r = torch.from_numpy(rep).to(dev)
    out = torch.empty((ncodes, 2 * nlag + 1, 2), dtype=torch.float64, device=dev)
    taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
    n_in = 70_001
    wide = np.clip(rng.normal(0, 3000, (n_in, 2)), -32768, 32767).astype(np.int16)
    h16 = frontend.fir_decimate(wide, taps, 14, out="int16")
    hf = frontend.fir_decimate(wide, taps, 14, out="f32")
    w = torch.from_numpy(wide).to(dev)
    y16 = torch.zeros((h16.shape[0], 2), dtype=torch.int16, device=dev)
    yf = torch.zeros((h16.shape[0], 2), dtype=torch.float32, device=dev)
    with Correlator(lfsr=(14, 43, 10000), fs=FS) as cor:
        for _ in range(2):                                             # second call reuses the context's buffers
            cor.sliding_dot_dev(x.data_ptr(), raw.shape[0], r.data_ptr(), nobs, ncodes, nlag, out.data_ptr(), pt=3, ff=2.5e-5, phi=0.125, scale=1 / 32768)
            nout = cor.fir_decimate_dev(w.data_ptr(), n_in, taps, 14, y16.data_ptr(), yf.data_ptr())
        cor.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[..., 0] + 1j * o[..., 1], host)
    assert nout == h16.shape[0] == (n_in - taps.size) // 14 + 1
    assert np.array_equal(y16.cpu().numpy(), h16)
    f = yf.cpu().numpy()
    assert np.array_equal((f[:, 0] + 1j * f[:, 1]).astype(np.complex64), hf)
    ref = orc.fir_decimate(wide[:, 0].astype(np.float64) + 1j * wide[:, 1], taps.astype(np.float64), 14)
    assert np.abs(hf - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-3


def test_cpp_twin_file_level_carrier_estimate(tmp_path):
    """GoRanging::df (processing/CPP/main.cpp:363-450): one carrier estimate per file from every 25th sample, an FFT of
    ARBITRARY length (here 39 999 and 40 003 records: odd, with prime factors far from 2/3/5) — Bluestein on the
    library's FFT against the oracle restatement (unpinned)."""
    from amaranth_twstft_amd import cpp_twin
    from tests.test_gpu_parity import _capture
    chips, raw = _capture(15, 3, 25000, 20, seed=5, df=(1780.75, -3.5))            # 1 000 000 samples x 2 channels
    for nsamp, foff in ((999_975 + 24, 0.0), (1_000_000, 250.0), (40_003 * 25, -1000.0)):
        r = np.concatenate([raw, raw])[:nsamp]
        path = tmp_path / f"cap{nsamp}.bin"
        r.tofile(path)
        got = cpp_twin.file_level_df(str(path), FS, 25, 0, foff)
        ref = orc.cpp_file_df(r, FS, 25, 0, foff)
        assert got == ref, (nsamp, got, ref)
        assert abs(got[0] - 1780.75) < 3.0 and abs(got[1] + 3.5) < 3.0               # 2.5 Hz bins (fs/25/nrec*... /2)
    eng = cpp_twin.ArbitraryFFT(1237)
    x = np.random.default_rng(2).normal(size=1237) + 1j * np.random.default_rng(3).normal(size=1237)
    assert np.abs(eng(x) - np.fft.fft(x)).max() <= 1e-10 * np.abs(np.fft.fft(x)).max()
    eng.close()


def test_file_df_behind_the_c_abi_and_the_goranging_program(tmp_path, monkeypatch):
    """twx_file_df (GoRanging::df, processing/CPP/main.cpp:363-450, on the device: Bluestein in blocks on the fp64 FFT of 5e6 points)
    against the oracle restatement and the Python twin, for series lengths that are odd / have large prime factors, with and without
    foffset, remote = 0 / 1, and with the block length forced small so that the block machinery (27 x 27 blocks for a 330-s capture)
    runs on a short one.  Then the program: apps/bin/goranging_hip writes the `<capture>C.mat` the Python twin writes (same variables,
    same values) and prints the program's lines."""
    from scipy.io import loadmat
    from amaranth_twstft_amd import cpp_twin, results_io
    from tests.test_gpu_parity import _capture
    lib = L.load()
    chips, raw = _capture(15, 3, 25000, 20, seed=5, df=(1780.75, -3.5))            # 1 000 000 samples x 2 channels
    cases = [(999_975 + 24, 0.0, 0, 0), (1_000_000, 250.0, 0, 0), (40_003 * 25, -1000.0, 0, 0), (1_000_000, 0.0, 1, 0),
             (1_000_000, 250.0, 0, 7001), (40_003 * 25, 0.0, 0, 12345), (999_999, 0.0, 0, 40000)]
    for nsamp, foff, remote, block in cases:
        r = np.concatenate([raw, raw])[:nsamp]
        path = tmp_path / f"cap{nsamp}_{block}.bin"
        r.tofile(path)
        if block:
            monkeypatch.setenv("TWX_FILEDF_BLOCK", str(block))
        else:
            monkeypatch.delenv("TWX_FILEDF_BLOCK", raising=False)
        d1, d2 = C.c_double(), C.c_double()
        rc = lib.twx_file_df(str(path).encode(), FS, 25, remote, foff, -1, C.byref(d1), C.byref(d2))
        assert rc == 0, lib.twx_file_df_last_error()
        ref = orc.cpp_file_df(r, FS, 25, remote, foff)
        assert d1.value == ref[0] and (np.isnan(d2.value) if remote else d2.value == ref[1]), (nsamp, foff, remote, block, d1.value, d2.value, ref)
    monkeypatch.delenv("TWX_FILEDF_BLOCK", raising=False)
    assert lib.twx_file_df(str(tmp_path / "nope.bin").encode(), FS, 25, 0, 0.0, -1, C.byref(d1), C.byref(d2)) != 0
    assert b"cannot open" in lib.twx_file_df_last_error()
    # ---- the program against the Python twin
    exe = os.path.join(ROOT, "apps", "bin", "goranging_hip")
    if not os.path.exists(exe):
        pytest.skip("apps/bin/goranging_hip not built")
    nchips, nwin = 25000, 6
    n = 2 * nchips
    chips, raw = _capture(15, 3, nchips, nwin, seed=11, df=(812.25, -3.5))
    (tmp_path / "run").mkdir()
    cap = tmp_path / "run" / "1670074501.bin"
    raw.tofile(cap)
    chips.tofile(tmp_path / "run" / "code.bin")
    out = subprocess.run([exe, str(cap), str(tmp_path / "run" / "code.bin")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert lines[0].endswith("data.bin code.bin [remote=0] [foffset=0.]") and f"{n} {3 * n}" in lines
    df1, df2 = cpp_twin.file_level_df(str(cap), FS, 25, 0, 0.0)
    assert "df1=%.3f" % df1 in lines and " df2=%.3f" % df2 in lines and "No more data" in lines and any(l.startswith("temps: ") for l in lines)
    rows = [l for l in lines if l.startswith(tuple(f"{p}/0 " for p in range(nwin)))]
    assert len(rows) == nwin and all(f"{p}/1 " in rows[p] for p in range(nwin))
    got = loadmat(str(tmp_path / "run" / "1670074501C.mat"))
    with Correlator(chips, fs=FS, Nint=1, window="hamming", var_ddof=0) as cor:
        r1 = cor.process_file(str(cap), n_channels=2, channel=0, df=df1)
        r2 = cor.process_file(str(cap), n_channels=2, channel=1, df=df2)
    (tmp_path / "twin").mkdir()
    want = loadmat(results_io.save_cpp_mat(str(tmp_path / "twin" / "1670074501.bin"), r1, r2))
    keys = [k for k in want if not k.startswith("__")]
    assert [k for k in got if not k.startswith("__")] == keys
    for k in keys:
        if k.startswith("SNR") or k.endswith("code"):              # 10*log10 in libm here, in numpy there: an ulp apart
            assert np.abs(got[k] - want[k]).max() <= 1e-13 * np.abs(want[k]).max(), k
        else:
            assert np.array_equal(got[k], want[k]), k
    d = float(rows[2].split()[1])
    assert abs(d - (r1[2].indice + r1[2].correction) / FS / 3) < 1e-11
    out = subprocess.run([exe, str(cap), str(tmp_path / "run" / "code.bin"), "1", "250"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and os.path.exists(tmp_path / "run" / "remote1670074501C.mat")
    g2 = loadmat(str(tmp_path / "run" / "remote1670074501C.mat"))
    assert "correction1" in g2 and "correction2" not in g2 and g2["df1"][0, 0] == cpp_twin.file_level_df(str(cap), FS, 25, 1, 250.0)[0]


def test_stream_contract_of_the_device_entry_point():
    """include/twstft_hip.h: twx_process_windows_dev is ordered against twx_stream(ctx) on both sides although batches run
    on several internal streams — a producer enqueued on that stream before the call (here the synthetic generator) and a
    consumer after it need no device-wide synchronisation.  Same records as the fully synchronised sequence, 40 windows
    = 5 batches over 3 slots, repeated with fresh data so that a race would have several chances to show."""
    import torch
    lib = L.load()
    dev = torch.device("cuda", 0)
    nchips, n, nwin = 100000, 200000, 40
    chips = chips_for(17, 9, nchips)
    cd = torch.from_numpy(chips).to(dev)
    band = L.twx_band(*band_godual(FS, n))
    with Correlator(chips, fs=FS, Nint=1, max_batch=8) as cor:
        st = lib.twx_stream(cor._h)
        for rep in range(4):
            iq = torch.zeros((nwin, n, 2), dtype=torch.int16, device=dev)
            res = torch.zeros((nwin, D.RESULT_BYTES), dtype=torch.uint8, device=dev)
            ref = torch.zeros((nwin, D.RESULT_BYTES), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            delays = []
            for w in range(nwin):
                d = 1000 + 37 * w + rep
                delays.append(d)
                p = synth.SynthParams(delay_q8=d * 256, fstep=synth.fstep_for_df(500.0 + w, FS), phi0=w, amp=300,
                                      noise_gain=synth.noise_gain_for_sigma(500.0), seed=900 + rep, stream=w)
                L.check(lib.twx_synth_capture_dev(iq[w].data_ptr(), n, 0, cd.data_ptr(), nchips, 2, 1, _params(p).ctypes.data_as(C.c_void_p), st))
            # producer still running on twx_stream(ctx): no synchronisation here
            L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
            host = np.empty((nwin, D.RESULT_BYTES), dtype=np.uint8)
            # consumer on the same stream: a plain stream synchronise of twx_stream(ctx) must be enough to see every record
            assert torch.cuda.ExternalStream(st).synchronize() is None
            L.check(lib.twx_memcpy_d2h(host.ctypes.data_as(C.c_void_p), res.data_ptr(), host.nbytes))
            torch.cuda.synchronize()
            L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, ref.data_ptr()), cor._h)
            cor.synchronize(); torch.cuda.synchronize()
            assert np.array_equal(host, ref.cpu().numpy())
            recs = D.results_from_bytes(host)
            assert [r.indice for r in recs] == [3 * d for d in delays]


def test_caf_bins_device_pointer_form():
    import torch
    dev = torch.device("cuda", 0)
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=6543 * 256, fstep=synth.fstep_for_df(7 * FS / n, FS), phi0=77, amp=400,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=8)
    raw = synth.synth_channel(n, chips, 2, p)
    iq = torch.from_numpy(raw).to(dev)
    with Correlator(chips, fs=FS, Nint=0) as cor:
        pk, lag = cor.caf_bins(raw, -40, 40)
        pkd, lagd = cor.caf_bins_dev(iq.data_ptr(), -40, 40)
    assert np.array_equal(lag, lagd) and np.array_equal(pk, pkd) and lag[47] == 6543


# --------------------------------------------------------------------------------------------------------------
# captures -> device tracked flow -> result files -> two-way combination -> <MJD>.1s  (acquisition/go_1s.m)
# --------------------------------------------------------------------------------------------------------------
def test_two_way_end_to_end_from_four_captures(tmp_path):
    """One session as the sites record it: OP and LTFB each capture their own loop-back (claudio_aligned_code_lo_separate.m)
    and the partner's signal ~50 kHz off (claudio_aligned_code_re_separate.m).  Product chain: four capture FILES ->
    twx_tracked_file (lo / re modes) -> the scripts' .mat records (gzip'ed, archive naming) -> twoway.process_sessions ->
    <MJD>.1s.  Oracle chain on the same captures: oracle.ranging_tracked -> oracle.go_1s_session (go_1s.m:77-268) ->
    oracle.go_1s_text.  Lags must agree exactly, the delivered 1-s delays to 0.02 ns (fp32 parabola vs fp64)."""
    import gzip
    from amaranth_twstft_amd import results_io, twoway
    from amaranth_twstft_amd.tracked import TrackedRanging
    nchips, n, ncodes = 10000, 20000, 155
    Lc = 50 * n
    codes = {"OP": prn.lfsr_chips(14, 43, nchips), "LTFB": prn.lfsr_chips(14, 57, nchips)}
    ts = {"OP": 1674402311, "LTFB": 1674402314}
    # (station, flavour) -> (code it correlates with, carrier Hz, delay samples, script OP flag, file name)
    plan = {("OP", "lo"): ("OP", 1234.5, 1500, 1, "OP/localclaudio%d_2" % ts["OP"]),
            ("OP", "re"): ("LTFB", -50007.0, 6400, 1, "OP/remoteclaudio%d_1" % ts["OP"]),
            ("LTFB", "lo"): ("LTFB", -777.25, 900, 0, "LTFB/localclaudio%d_1" % ts["LTFB"]),
            ("LTFB", "re"): ("OP", 49991.0, 7100, 0, "LTFB/remoteclaudio%d_2" % ts["LTFB"])}
    (tmp_path / "OP").mkdir(); (tmp_path / "LTFB").mkdir()
    oracle_rec = {}
    for i, ((station, flavour), (code_of, car, delay, OP, name)) in enumerate(plan.items()):
        chips = codes[code_of]
        p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(car, FS), phi0=7 + i, amp=900,
                              noise_gain=synth.noise_gain_for_sigma(120.0), seed=400 + i)
        raw = synth.synth_channel(n * ncodes, chips, 2, p)
        cap = tmp_path / f"cap_{station}_{flavour}.bin"
        raw.tofile(cap)
        m = orc.tracked_mode(flavour, OP)
        with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc, mode=flavour, OP=OP) as tr:
            got = tr.run_file(str(cap), skip_seconds=0.0)
        want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, band=m["band"], carrier=m["carrier"], indice_floor=m["indice_floor"])
        assert got["indice1"] == want["indice1"] and len(want["indice1"]) >= 148 and got["df"] == want["df"], (station, flavour)
        mat = tmp_path / (name + ".mat")
        results_io.save_tracked_mat(str(mat), got, code=orc.make_code(chips, 2))
        with open(mat, "rb") as f, gzip.open(str(mat) + ".gz", "wb") as g:
            g.write(f.read())
        mat.unlink()
        oracle_rec[(station, flavour)] = {k: np.asarray(want[k2]) for k, k2 in (("xval1", "xval"), ("indice1", "indice1"), ("correction1", "correction1"),
                                                                                 ("SNR1r", "SNR1r"), ("SNR1i", "SNR1i"))}
    out = twoway.process_sessions(str(tmp_path), out_dir=str(tmp_path))
    assert len(out) == 1
    mjd, tw, path = out[0]
    want = orc.go_1s_session(oracle_rec[("OP", "lo")], oracle_rec[("OP", "re")], oracle_rec[("LTFB", "lo")], oracle_rec[("LTFB", "re")])
    assert want is not None and want["rows"].shape[0] >= 4 and len(want["oplo"]) > 102
    assert len(tw.oplo) == len(want["oplo"]) and tw.one_second.shape == want["rows"].shape
    got_lines = open(path).read().split("\n")
    want_lines = orc.go_1s_text(twoway.mjd_of_unix(ts["LTFB"]), want["rows"]).split("\n")
    assert got_lines[0] == want_lines[0] and len(got_lines) == len(want_lines)
    for a, b in zip(got_lines[1:], want_lines[1:]):
        if not b:
            continue
        ga, wb = [float(v) for v in a.split("\t")], [float(v) for v in b.split("\t")]
        assert ga[0] == wb[0] and max(abs(x - y) for x, y in zip(ga[1:], wb[1:])) <= 0.02, (a, b)
    # the two-way observable itself: res (ns), NaN pattern included
    res_w = want["res"]                                            # incl. the shift of :210-211: the product applies the script's lines by default (round 5)
    assert np.array_equal(np.isnan(tw.res), np.isnan(res_w)) and np.nanmax(np.abs(tw.res - res_w)) <= 0.03
    # every channel was re-aligned to sample 21 by the tracked loop (:183), so the four series sit at 21 samples = 4200 ns
    for series in (tw.oplo, tw.opre, tw.ltlo, tw.ltre):
        assert np.abs(series - 21 / FS * 1e9).max() < 70.0          # the remote series keep indice/3 = 21 1/3 (:174)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_full_size_carrier_search_and_map_at_row_boundaries(precision):
    """N = 5e6 = 625 x 8000 (the folded, row-walking k_rowd in fp32; the folded k_rowd in fp64): (a) the coarse carrier estimate finds
    a tone whose square lies on the first / last bins of the band, next to zero and on both sides of the k2 = 0 / 7999 row ends of the
    two-pass layout; (b) the correlation peak placed on the last and the first element of a row (lag = 8000 m - 1, 8000 m) and on the
    map's ends comes out with the oracle's lag, neighbours and correction."""
    import torch
    dev = torch.device("cuda", 0)
    chips = chips_for(22, 3, NCHIPS)
    freq = orc.freq_axis(FS, N)
    k = orc.band_godual(freq)
    band = band_godual(FS, N)
    # (a) tones: the replica of the generator held at -1 (chips all zero) is a pure carrier
    kappas = [int(k[0]) - N // 2, int(k[-1]) - N // 2, -1, 1, 624, 625, 626, -625, 7999, 8000, 8001, -8001, 15999, 16000, 19999, -19999]
    zeros = torch.zeros(NCHIPS, dtype=torch.uint8, device=dev)
    iq = torch.empty((N, 2), dtype=torch.int16, device=dev)
    with Correlator(chips, fs=FS, Nint=1, precision=precision, max_batch=1) as cor:
        for kappa in kappas:
            p = synth.SynthParams(delay_q8=0, fstep=synth.fstep_for_df(kappa / 2.0 * FS / N, FS), phi0=12345, amp=6000, noise_gain=0, seed=1)
            _synth_dev(iq, N, zeros, NCHIPS, 2, [p])
            torch.cuda.synchronize()
            g = cor.process_dev(iq.data_ptr(), 1, 1, 0, band=band)[0]
            d = orc.deinterleave(iq.cpu().numpy(), 1, 0)
            idx, df = orc.coarse_df(d - d.mean(), k, freq)                     # (the mean removal bends the tones next to zero: ask the oracle)
            assert g.df_index == idx and abs(g.df - df) <= 1e-9, (kappa, g.df_index - N // 2, idx - N // 2)
            assert abs(idx - N // 2 - kappa) <= 1
        # (b) peaks across the row ends
        code = orc.make_code(chips, 2)
        fcode = orc.make_fcode(code)
        temps = np.arange(N) / FS
        cd = torch.from_numpy(chips).to(dev)
        for delay in (8000 * 311 - 1, 8000 * 311, N - 1, 0):
            p = synth.SynthParams(delay_q8=delay * 256 + 60, fstep=synth.fstep_for_df(333.0, FS), phi0=7, amp=300, noise_gain=synth.noise_gain_for_sigma(400.0), seed=delay + 5)
            _synth_dev(iq, N, cd, NCHIPS, 2, [p])
            torch.cuda.synchronize()
            raw = iq.cpu().numpy()
            g = cor.process(raw, 1, 0, df=333.0)[0]
            d = orc.deinterleave(raw, 1, 0)
            d = d - d.mean()
            o = orc.processing(d, None, None, temps, fcode, code, Nint=1, fs=FS, df=333.0)
            assert g.indice == o["indice"], (delay, g.indice, o["indice"])
            off = (g.indice / 3.0 - delay) % N
            assert min(off, N - off) <= 1.5, (delay, g.indice)                 # the generator's truth, circularly
            tol = (2e-6 if precision == "f32" else 1e-11) * abs(o["xval"])
            assert abs(g.xval - o["xval"]) <= tol and abs(g.xvalm1 - o["xvalm1"]) <= tol and abs(g.xvalp1 - o["xvalp1"]) <= tol, delay
            assert abs(g.correction - o["correction"]) <= 2e-4 and abs(g.SNRr - o["SNRr"]) <= 3e-4 * max(o["SNRr"], o["SNRi"])


def test_randomised_acquisition_sweeps():
    """The one-call acquisition (twx_acquire_cdev: coarse sweep, step halving until < 1 Hz, bookkeeping on the device) with random
    centre, range and step — few and many trial carriers per round, the best carrier inside the range, on its edge or absent, the
    B210 build's decimated sweep — against orc.rx_acquire on the same x2-interpolated stream (half rate: fs_in = 2.5 Msps, nobs =
    200 000, nfft = 2^19 / 2^18): integer carrier and cblas_izamax lag exact, the peak within 3e-6.  TWX_SWEEP_OPTIONS raises the count."""
    import torch
    from amaranth_twstft_amd import acquisition as acq
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8128 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "4"))
    n_in, fs, rc, clen = 2_500_000, 5e6, 2.5e6, 100_000
    nobs = int(fs) // 25
    chips = chips_for(17, 9, clen)
    code_pm1 = 1 - 2 * chips.astype(np.int64)
    interp = acq.Interpolator(n_in)
    smp_dev = torch.empty((2 * n_in, 2), dtype=torch.float32, device=dev)
    accs = {d: acq.Acquisition(code_pm1, rc, fs, nobs, dec_a=d) for d in (1, 2)}
    wavs = {d: orc.rx_replica(code_pm1, nobs, accs[d].nfft, rc, fs, clen, rc, -rc, dec_a=d)[0] for d in (1, 2)}
    try:
        for it in range(ncomb):
            fc_true = float(rng.uniform(-3000, 3000))
            amp = int(rng.choice([0, 500, 1500]))
            p = synth.SynthParams(delay_q8=int(rng.integers(0, clen)) * 256, fstep=synth.fstep_for_df(fc_true, 2.5e6), phi0=int(rng.integers(0, 1 << 30)), amp=amp,
                                  noise_gain=synth.noise_gain_for_sigma(700.0), seed=int(rng.integers(1, 10 ** 6)))
            raw1 = synth.synth_channel(n_in, chips, 1, p)
            raw = np.zeros((n_in, 4), dtype=np.int16)
            raw[:, 0:2] = raw1
            iq = torch.from_numpy(raw).to(dev)
            interp(iq.data_ptr(), smp_dev.data_ptr(), n_channels=2, channel=0)
            interp.cor.synchronize()
            oA, _ = orc.rx_short2double(raw.reshape(-1), 2 * n_in)
            dec_a = int(rng.integers(1, 3))
            a = accs[dec_a]
            fstep = float(rng.choice([16.0, 64.0, 256.0]))
            frange = fstep * float(rng.choice([2, 4, 16]))
            fc_init = float(np.round(fc_true + rng.choice([0.0, 0.4, -0.9, 1.0, 3.0]) * frange))      # inside, at the edge, outside the range
            idx = int(rng.integers(0, (2 * n_in - a.nfft * dec_a) // nobs)) * nobs
            tag = f"combination {it}: dec_a={dec_a} fc_true={fc_true:.1f} amp={amp} fc_init={fc_init} range={frange} step={fstep} idx={idx}"
            fc, pk, pt = a.acquire(smp_dev.data_ptr(), idx, fc_init=fc_init, frange=frange, fstep=fstep)
            fo, pko, pto = orc.rx_acquire(oA, idx, wavs[dec_a], nobs, a.nfft, fs, fc_init, frange, fstep, rc, -rc, dec_a=dec_a)
            assert (fc, pt) == (fo, pto), (tag, fc, fo, pt, pto)
            assert abs(pk - pko) <= 3e-6 * pko, tag
    finally:
        for a in accs.values():
            a.close()
        interp.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_early_prompt_late_tracking_loop(precision):
    """experiments/230503_100kchips_withcode/gotracking_inv2.m:149-235 — the Octave DLL/PLL experiment's main loop: NCO mix, three LINEAR
    cross-correlations over every lag (xcorr(al|ap|ae, xx, MAXLAG = points_per_code)) on the GPU, early-minus-late and arctangent
    discriminators, 2nd-order loop filter — 25 code periods of a 5000-chip code with a residual carrier, against oracle.epl_step block by
    block (peak indices equal, peak values and every loop quantity to the precision of the context)."""
    from amaranth_twstft_amd import epl
    chips = chips_for(13, 27, 5000)
    fs, n = FS, 10000
    al, ap, ae = epl.replicas(chips, 2)
    rng = np.random.default_rng(8)
    nblk, delay, f_res, freq0 = 25, 3, 4.0, 1000.0
    t = np.arange(nblk * n + n) / fs
    sig = 800.0 * np.tile(ap, nblk + 1) * np.exp(2j * np.pi * ((freq0 + f_res) * t + 0.11))
    sig = np.roll(sig, delay) + rng.normal(0, 300.0, sig.size) + 1j * rng.normal(0, 300.0, sig.size)
    sig = np.rint(sig.real) + 1j * np.rint(sig.imag)                                    # what an int16 capture holds
    state = dict(l=1, doppler_freq=[0.0], time_end=-1.0 / fs, code_phase=0.0, carrier_phase=0.0)
    tol = 5e-9 if precision == "f64" else 2e-5
    with epl.EplTracker(chips, fs=fs, freq0=freq0, time_end=-1.0 / fs, precision=precision) as trk:
        for b in range(nblk):
            x = sig[b * n:(b + 1) * n]
            got = trk.step(x)
            want = orc.epl_step(state, x, al, ap, ae, fs=fs, freq0=freq0)
            assert (got["bbl"], got["bbp"], got["bbe"]) == (want["bbl"], want["bbp"], want["bbe"]), (b, got["bbp"], want["bbp"])
            for k in ("zl", "zp", "ze"):
                assert abs(got[k] - want[k]) <= tol * abs(want[k]), (b, k, got[k], want[k])
            for k in ("code_phase_error", "delta_theta", "sortie", "filtered_code_phase", "filtered_carrier_phase", "measured_doppler_freq", "doppler_freq"):
                assert abs(got[k] - want[k]) <= 50 * tol * max(1.0, abs(want[k])), (b, k, got[k], want[k])
            # the loop feeds its own estimate back (a 1e-12 difference in delta_theta is a different NCO in the next block): keep the twins in step
            if True:
                trk.doppler_freq[-1], trk.code_phase, trk.carrier_phase = state["doppler_freq"][-1], state["code_phase"], state["carrier_phase"]
        assert got["bbp"] == n + 1 - delay                                                      # xcorr(ap, xx) peaks at lag -delay (index lag + N + 1)
        assert got["bbl"] - got["bbp"] == 1 and got["bbp"] - got["bbe"] == 1                  # late / early replicas: one sample either side
        assert len(trk.doppler_freq) == nblk + 1 and trk.l == nblk + 1
        # (with the script's constants — T_blk = 80 ms, B_PLL = 20 Hz, made for 40-ms codes — a 2-ms code period does not pull in; the
        # test holds the twin against the restatement, not the loop design)


def test_code_stepping_tracking_loop_on_the_sliding_dot_product():
    """experiments/230503_100kchips_withcode/gotracking_test.m:121-187 — the SECOND DLL experiment: after the first block (every lag) the three
    correlations run over +-20 lags only, i.e. on the direct sliding dot product of the tracking stage (twx_sliding_dot, NCO inside the kernel);
    the coherent discriminator d steps the code when it leaves (-0.5, 0.5).  A 5000-chip code whose delay DRIFTS by a sample every few
    blocks, against oracle.codestep_step block by block: peak indices, the three peak values, d, the steps taken, the loop's frequency."""
    from amaranth_twstft_amd import epl
    chips = chips_for(13, 27, 5000)
    fs, n = FS, 10000
    _, ap, _ = epl.replicas(chips, 2)
    rng = np.random.default_rng(18)
    nblk, freq0, f_res = 30, 1000.0, 0.7
    total = (nblk + 1) * n
    t = np.arange(total) / fs
    # The three correlations each find their OWN peak, so d stays near zero while all three peaks are inside the +-20 lags; it leaves
    # (-0.5, 0.5) — and the code is stepped — when the received code has slid so far that the early (or late) replica's peak falls outside
    # the window and only a sidelobe is left of it.  The received code slides one sample every 3 blocks, starting 17 samples off.
    delay = 17 + (np.arange(total) // (3 * n))
    idx = (np.arange(total) - delay) % n
    sig = 900.0 * ap[idx] * np.exp(2j * np.pi * ((freq0 + f_res) * t + 0.07)) + rng.normal(0, 250.0, total) + 1j * rng.normal(0, 250.0, total)
    raw = np.empty(2 * total, dtype=np.int16)
    raw[0::2] = np.rint(sig.real); raw[1::2] = np.rint(sig.imag)
    al0, ap0, ae0 = epl.replicas(chips, 2)
    state = dict(l=1, maxlag=n, ap=ap0, al=al0, ae=ae0, freq=0.0, freqm1=0.0, freqm2=0.0, ym1=0.0, time_end=-1.0 / fs)
    trk = epl.CodeStepTracker(chips, fs=fs, freq0=freq0, time_end=-1.0 / fs)
    steps = 0
    for b in range(nblk):
        blk = raw[2 * b * n: 2 * (b + 1) * n]
        x = blk[0::2].astype(float) + 1j * blk[1::2].astype(float)
        got = trk.step(blk)
        want = orc.codestep_step(state, x, fs=fs, freq0=freq0)
        tol = 5e-9 if b == 0 else 3e-6                           # the first block on the fp64 contexts, the others on the fp32 sliding dot product
        assert (got["bbl"], got["bbp"], got["bbe"]) == (want["bbl"], want["bbp"], want["bbe"]), (b, got["bbp"], want["bbp"])
        for k in ("zl", "zp", "ze"):
            assert abs(got[k] - want[k]) <= tol * abs(want[k]), (b, k, got[k], want[k])
        assert abs(got["d"] - want["d"]) < 1e-4 and got["stepped"] == want["stepped"], (b, got["d"], want["d"])
        assert abs(got["yp"] - want["yp"]) < 1e-4 and abs(got["freq"] - want["freq"]) < 1e-5, (b, got["freq"], want["freq"])
        trk.freq, trk.freqm1, trk.freqm2, trk.ym1 = state["freq"], state["freqm1"], state["freqm2"], state["ym1"]      # keep the twins in step (the loop feeds back)
        assert np.array_equal(trk.ap, state["ap"])
        steps += abs(got["stepped"])
    assert steps >= 3 and trk.maxlag == 20                      # the code was stepped as the peak reached the edge of the lag window
