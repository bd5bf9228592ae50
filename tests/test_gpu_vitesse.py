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
