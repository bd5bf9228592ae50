"""GPU (-m gpu): the velocity-compensated window of experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m — after the NCO mix the window
is resampled linearly on a stretched axis (``interp1`` :40), the offset ``t0`` carried from window to window (:41) with its wrap into
(-1, 1) and the whole-sample count ``dt`` (:68-71), the NaN edge rule of :42-43 — against the numpy restatement
``oracle.ranging_vitesse`` (UNPINNED: the script is Octave only and the reference holds no output of it).
Gates as everywhere: integer lag bit-exact, |peak| within 1e-6 relative."""
import numpy as np
import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, freq_axis
from oracle import twstft_oracle as orc

pytestmark = pytest.mark.gpu
FS = 5e6


def _band(n, lo=96200.0, hi=106200.0):
    f = freq_axis(FS, n)
    k = np.nonzero((f < hi) & (f > lo))[0]
    return int(k[0]), int(k[-1])


def _capture(chips, nwin, df=50130.0, seed=40):
    n = 2 * len(chips)
    ps = [synth.SynthParams(delay_q8=(777 + 3 * w) * 256 + 90, fstep=synth.fstep_for_df(df, FS), phi0=w * 977, amp=300,
                            noise_gain=synth.noise_gain_for_sigma(250.0), seed=seed + w) for w in range(nwin)]
    return np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps]).reshape(-1)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("nchips,bitlen,taps,vitesse,nwin", [
    (2500, 13, 27, -7e-5, 9),        # t0 = 0, -0.35, -0.7, wrap ... dt climbs; the first sample's query leaves the window (edge rule)
    (2500, 13, 27, +6e-5, 9),        # the other sign: the LAST sample's query leaves the window, dt falls — and with t0 near one MORE than
                                     # the last sample does: the script's map is NaN there (interp1 extrapolates nothing), and so is the record
    (10000, 14, 43, -3.25e-6, 5),    # the script's sign, scaled to a 4-ms window
    (250000, 22, 3, -3.25e-9, 3),    # the script's own value on a 100-ms window
])
def test_velocity_compensated_windows_match_the_script(precision, nchips, bitlen, taps, vitesse, nwin):
    """Multi-window captures where ``t0`` accumulates and wraps: per window the lag is bit-exact, the three peak samples within 1e-6 of
    the peak, the 3-point polyfit vertex (:56-57) within 1e-5 sample, ``df`` equal, ``dt`` and the carried state equal to the script's;
    in two calls (state carried across calls) as in one."""
    chips = prn.lfsr_chips(bitlen, taps, nchips)
    n = 2 * nchips
    raw = _capture(chips, nwin)
    band = _band(n)
    want, t0_end, dt_end = orc.ranging_vitesse(raw, chips, fs=FS, vitesse=vitesse, n_channels=1, channel=0)
    assert any(w["nan"] for w in want) == (vitesse > 0)
    with Correlator(chips, fs=FS, Nint=0, precision=precision, code_levels="unipolar", code_zero_mean=True) as cor:
        plain = cor.process(raw, n_channels=1, channel=0, band=band)
        cor.set_resample(vitesse)
        got = cor.process(raw, n_channels=1, channel=0, band=band)
        v, t0, dt = cor.get_resample()
        assert v == vitesse and dt == dt_end and abs(t0 - t0_end) < 1e-12
        # the same in two calls: the state carries over
        cor.set_resample(vitesse, 0.0, 0)
        k = nwin // 2
        two = cor.process(raw[:k * n * 2], n_channels=1, channel=0, band=band) + cor.process(raw[k * n * 2:], n_channels=1, channel=0, band=band)
        cor.set_resample(0.0)
        off = cor.process(raw, n_channels=1, channel=0, band=band)
    tol = 1e-6 if precision == "f32" else 1e-9
    moved = 0
    for w, (g, o) in enumerate(zip(got, want)):
        if o["nan"]:
            assert g.status & L.TWX_STATUS_RESAMPLE_NAN and g.indice == 0 and g.dt == o["dt"] and np.isnan(g.xval.real) and np.isnan(g.correction)
            moved += 1
            continue
        assert g.indice == o["indice"] and g.dt == o["dt"] and g.status == 0, (w, g.indice, o["indice"], g.dt, o["dt"])
        pk = abs(o["xval"])
        assert abs(g.xval - o["xval"]) <= tol * pk and abs(g.xvalm1 - o["xvalm1"]) <= tol * pk and abs(g.xvalp1 - o["xvalp1"]) <= tol * pk
        assert abs(g.correction_polyfit(1) - o["correction"]) < 1e-5 and abs(g.correction - o["correction"]) < 1e-5     # the closed form IS the 3-point fit
        assert abs(g.df - o["df"]) < 1e-9
        assert abs((g.indice + 1 + g.dt + g.correction) - o["solution"]) < 1e-5
        moved += int(abs(g.xval - plain[w].xval) > 1e-4 * pk)
    assert moved >= nwin - 1                                              # the option changes the map (all but the t0 = 0 window at the least)
    same = lambda a, b: (a.indice, a.dt, a.status) == (b.indice, b.dt, b.status) and (a.xval == b.xval or (np.isnan(a.xval.real) and np.isnan(b.xval.real)))
    assert all(same(a, b) for a, b in zip(two, got))
    assert [(a.indice, a.xval, a.dt) for a in off] == [(a.indice, a.xval, 0) for a in plain]


def test_velocity_window_device_and_file_entry_points(tmp_path):
    """The same records through twx_process_windows_dev (device-resident capture, several batches) and twx_process_file (pinned ingest
    pipeline, skip): the carried state moves by the windows actually processed."""
    import torch
    chips = prn.lfsr_chips(13, 27, 2500)
    n, nwin, v = 5000, 21, -4e-5
    raw = _capture(chips, nwin, seed=90)
    band = _band(n)
    want, t0_end, dt_end = orc.ranging_vitesse(raw, chips, fs=FS, vitesse=v, n_channels=1, channel=0)
    path = tmp_path / "cap.bin"
    raw.tofile(path)
    with Correlator(chips, fs=FS, Nint=0, code_levels="unipolar", code_zero_mean=True, max_batch=4) as cor:
        cor.set_resample(v)
        dev = torch.from_numpy(raw).cuda()
        got = cor.process_dev(dev.data_ptr(), nwin, band=band)
        assert [(g.indice, g.dt) for g in got] == [(o["indice"], o["dt"]) for o in want]
        assert cor.get_resample()[2] == dt_end
        cor.set_resample(v, 0.0, 0)
        gotf = cor.process_file(str(path), n_channels=1, channel=0, band=band)
        assert [(g.indice, g.dt, g.xval) for g in gotf] == [(g.indice, g.dt, g.xval) for g in got]
        assert abs(cor.get_resample()[1] - t0_end) < 1e-12 and cor.get_resample()[2] == dt_end
        # the whole-window NaN case: |t0| so large that more than the edge sample leaves the window — the script's map is NaN there
        cor.set_resample(v, -1.5, 0)
        bad = cor.process(raw[:2 * n], n_channels=1, channel=0, band=band)
        wantb, _, _ = orc.ranging_vitesse(raw[:2 * n], chips, fs=FS, vitesse=v, n_channels=1, channel=0, t0=-1.5)
        assert wantb[0]["nan"] and bad[0].status & L.TWX_STATUS_RESAMPLE_NAN and bad[0].indice == 0 and np.isnan(bad[0].xval.real)
        with pytest.raises(L.TwxError):
            cor.process(np.zeros(4 * n, dtype=np.int16), n_channels=2, channel=-1, band=band)      # one channel at a time
        with pytest.raises(L.TwxError):
            cor.set_resample(1e-3)                                                                   # a sample or more per window


def test_mex_gateway_option_forms_carry_the_velocity_window(tmp_path):
    """Call forms D / E of mex/twstft_processing_mex.cpp executed on the functional fake mex.h — what mex/godual_ranging_OP_vitesse_hip.m
    runs on: 'option' 'replica' / 'vitesse' / 'snr_estimators' before a raw-int16 call, then 'extra' (bruit, valmax_square, noise_square,
    status, dt per record) and the carried state — against the same windows through ctypes."""
    import os
    import subprocess
    from tests.test_abi_and_host import build_mex_harness
    from tests.test_gpu_configs import _read_mex_outputs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = build_mex_harness(root, tmp_path)
    chips = prn.lfsr_chips(13, 27, 2500)
    n, nwin, v = 5000, 7, -7e-5
    raw = _capture(chips, nwin)
    raw.tofile(tmp_path / "cap.bin")
    chips.tofile(tmp_path / "chips.bin")
    band = _band(n)
    r = subprocess.run([str(exe), "raw", str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"), "1", "1",
                        str(band[0] + 1), str(band[1] + 1), "5e6", "0", "opt=replica:unipolar_zero_mean", "opt=vitesse:%r,0,0" % v,
                        "opt=snr_estimators:101,301", "extra=1", "state=1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    o = _read_mex_outputs(tmp_path / "out.bin")
    assert len(o) == 15
    with Correlator(chips, fs=FS, Nint=0, code_levels="unipolar", code_zero_mean=True, var_ddof=1) as cor:
        cor.set_resample(v)
        cor.set_snr_estimators(101, 301)
        want = cor.process(raw, n_channels=1, channel=0, band=band)
        ex = cor.snr_estimators(nwin)
        state = cor.get_resample()
    for w, g in enumerate(want):
        assert o[0][0, w] == g.indice + 1 and o[8][0, w] == g.xval and o[4][0, w] == g.df and o[1][0, w] == g.correction
        assert o[12][0, w] == g.status and o[13][0, w] == g.dt
        for i, key in enumerate(("bruit", "valmax_square", "noise_square")):
            assert o[9 + i][0, w] == ex[w][key] or (np.isnan(o[9 + i][0, w]) and np.isnan(ex[w][key]))
    assert tuple(o[14][0]) == (state[0], state[1], float(state[2])) and any(g.dt for g in want)


def test_script_level_mirror_of_the_vitesse_loop(tmp_path):
    """amaranth_twstft_amd/vitesse.py: the two-channel capture loop of the script (channel 1 resampled, channel 2 plain) against the oracle
    on both channels: ``solution12 - solution22`` equal to 1e-5 sample."""
    from amaranth_twstft_amd import vitesse
    chips = prn.lfsr_chips(13, 27, 2500)
    n, nwin, v = 5000, 6, -5e-5
    ps1 = [synth.SynthParams(delay_q8=(900 + w) * 256, fstep=synth.fstep_for_df(50130.0, FS), phi0=w, amp=300, noise_gain=synth.noise_gain_for_sigma(250.0), seed=7 + w) for w in range(nwin)]
    ps2 = [synth.SynthParams(delay_q8=200 * 256, fstep=0, phi0=w, amp=2000, noise_gain=synth.noise_gain_for_sigma(100.0), seed=70 + w, stream=1) for w in range(nwin)]
    raw = np.concatenate([synth.synth_capture(n, chips, 2, [a, b]) for a, b in zip(ps1, ps2)]).reshape(-1)
    path = tmp_path / "OP.bin"
    raw.tofile(path)
    got = vitesse.ranging_vitesse(str(path), chips, fs=FS, vitesse=v)
    want1, _, _ = orc.ranging_vitesse(raw, chips, fs=FS, vitesse=v, n_channels=2, channel=0)
    code = orc.make_code_variant(chips, unipolar=True, zero_mean=True)
    fcode = np.conj(np.fft.fft(code))
    for w in range(nwin):
        d2 = orc.deinterleave(raw[w * n * 4:(w + 1) * n * 4], 2, 1)
        d2 = d2 - d2.mean()
        pm = np.fft.ifft(np.fft.fft(d2) * fcode)
        i2, c2, x2, _, _ = orc.peak_refine(pm)
        assert got["indice2"][w] == i2 + 1 and abs(got["correction22"][w] - c2) < 1e-5 and abs(got["xval2"][w] - x2) <= 1e-6 * abs(x2)
        assert got["indice1"][w] == want1[w]["indice"] + 1 + want1[w]["dt"] and abs(got["solution12"][w] - want1[w]["solution"]) < 1e-5
    assert got["delay"].shape == (nwin,) and np.all(got["status"] == 0)
