"""GPU (-m gpu): the other SNR estimators the reference compares — experiments/220830_OP/process_OP.m:94-97,119-121,138 (``bruit``,
``valmax_square``, ``noise_square``) and the three-estimator comparison of experiments/221127_SNR/simu_snr.m / README.md — as optional
outputs of the device path, against the numpy restatements ``oracle.snr_offpeak`` / ``oracle.snr_square`` (UNPINNED: Octave only)."""
import numpy as np
import pytest

from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd.correlator import Correlator, band_godual
from oracle import twstft_oracle as orc

pytestmark = pytest.mark.gpu
FS = 5e6


def _oracle(raw, chips, n, band, Nint, nwin, bruit_len, sq_len, convention="godual"):
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code, "godual")
    freq = orc.freq_axis(FS, n)
    k = np.arange(band[0], band[1] + 1)
    temps = np.arange(n) / FS
    out = []
    for w in range(nwin):
        d = orc.deinterleave(raw[w * n * 2:(w + 1) * n * 2], 1, 0)
        d = d - d.mean()
        vmax, noise, tmp = orc.snr_square(d, k, sq_len)
        df = freq[tmp] / 2
        y = d * np.exp(-1j * 2 * np.pi * df * temps)
        prnmap = orc.xcorr_interp(orc._fft(y), fcode, Nint)
        if convention == "claudio":
            prnmap = np.conj(np.roll(prnmap[::-1], 1))                 # prnmap_c[m] = conj(prnmap_g[(M - m) mod M])
        ind = int(np.abs(prnmap).argmax())
        out.append(dict(indice=ind, bruit=orc.snr_offpeak(prnmap, ind, bruit_len), valmax_square=vmax, noise_square=noise))
    return out


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("nchips,bitlen,taps,Nint,bruit_len,sq_len,convention", [
    (100000, 17, 9, 1, 1001, 10001, "godual"),      # the 40-ms code of the tracked flow: rows of 400 points (k_rowd_small), ranges as in the script
    (250000, 22, 3, 1, 10001, 10001, "godual"),     # 100-ms window, bruit2's length
    (250000, 22, 3, 0, 1001, 4001, "claudio"),      # no interpolation, the mirrored map of the claudio convention
    (10000, 14, 43, 1, 1001, 1001, "godual"),       # short window: the ranges wrap / leave the map for late peaks -> NaN as the guard says
])
def test_offpeak_and_squared_spectrum_estimators_match_the_script(precision, nchips, bitlen, taps, Nint, bruit_len, sq_len, convention):
    chips = prn.lfsr_chips(bitlen, taps, nchips)
    n, nwin = 2 * nchips, 5
    delays = [1234, n // 2, n - 300, 77, n - 7000]                     # peaks early, in the middle and so late that indice+20+L leaves the map
    ps = [synth.SynthParams(delay_q8=d * 256, fstep=synth.fstep_for_df(1500.0 + 100 * w, FS), phi0=w, amp=250, noise_gain=synth.noise_gain_for_sigma(300.0), seed=60 + w)
          for w, d in enumerate(delays)]
    raw = np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps]).reshape(-1)
    band = band_godual(FS, n)
    want = _oracle(raw, chips, n, band, Nint, nwin, bruit_len, sq_len, convention)
    with Correlator(chips, fs=FS, Nint=Nint, precision=precision, convention=convention, var_ddof=1) as cor:
        base = cor.process(raw, n_channels=1, channel=0, band=band)
        cor.set_snr_estimators(bruit_len, sq_len)
        got = cor.process(raw, n_channels=1, channel=0, band=band)
        ex = cor.snr_estimators(nwin)
        assert [(g.indice, g.xval, g.SNRr) for g in got] == [(g.indice, g.xval, g.SNRr) for g in base]      # the records themselves do not change
        tol = 2e-5 if precision == "f32" else 1e-9
        seen_nan = 0
        for w in range(nwin):
            assert got[w].indice == want[w]["indice"]
            for key in ("bruit", "valmax_square", "noise_square"):
                a, b = ex[w][key], want[w][key]
                if np.isnan(b):
                    assert np.isnan(a), (w, key, a)
                    seen_nan += 1
                else:
                    assert abs(a - b) <= tol * abs(b), (w, key, a, b)
        assert seen_nan >= 1                                            # the guard of process_OP.m:119 was exercised
        # df supplied: no squared spectrum is formed
        cor.process(raw, n_channels=1, channel=0, df=[g.df for g in got])
        ex2 = cor.snr_estimators(nwin)
        assert all(np.isnan(e["valmax_square"]) and np.isnan(e["noise_square"]) for e in ex2)
        assert all((np.isnan(a["bruit"]) and np.isnan(b["bruit"])) or abs(a["bruit"] - b["bruit"]) <= 1e-6 * abs(b["bruit"]) for a, b in zip(ex2, ex))
        cor.set_snr_estimators(0, 0)
        from amaranth_twstft_amd import _lib as L
        with pytest.raises(L.TwxError):
            cor.process(raw, n_channels=1, channel=0, band=band); cor.snr_estimators(nwin)


def test_full_size_window_estimators_and_device_entry():
    """The 1-s window of BASELINE.json configs[1] (rows of 8000 points, 625 rows): the script's own lengths (1001 / 10001) through the
    device-resident entry, several batches, against the oracle on two of the windows."""
    import torch
    nchips = 2_500_000
    chips = prn.lfsr_chips(22, 3, nchips)
    n, nwin = 2 * nchips, 10
    ps = [synth.SynthParams(delay_q8=(1311765 + 11 * w) * 256, fstep=synth.fstep_for_df(1780.75 + w, FS), phi0=w, amp=200, noise_gain=synth.noise_gain_for_sigma(400.0), seed=7 + w)
          for w in range(nwin)]
    raw = np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps]).reshape(-1)
    band = band_godual(FS, n)
    dev = torch.from_numpy(raw).cuda()
    with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as cor:
        cor.set_snr_estimators(1001, 10001)
        got = cor.process_dev(dev.data_ptr(), nwin, band=band)
        ex = cor.snr_estimators(nwin)
    assert [g.indice for g in got] == [3 * (1311765 + 11 * w) for w in range(nwin)]
    for w in (0, 9):
        want = _oracle(raw[w * n * 2:(w + 1) * n * 2], chips, n, band, 1, 1, 1001, 10001)[0]
        for key in ("bruit", "valmax_square", "noise_square"):
            assert abs(ex[w][key] - want[key]) <= 2e-5 * abs(want[key]), (w, key, ex[w][key], want[key])
    assert all(np.isfinite([e["bruit"], e["valmax_square"], e["noise_square"]]).all() for e in ex)


def test_three_estimators_behave_as_the_reference_says():
    """experiments/221127_SNR/simu_snr.m + README.md on seeded synthetic captures: with the noise fixed and the signal raised over four
    decades of power, (i) the wipe-off estimate (``SNRclaudio = mean(x.*signal)^2/var(x.*signal)``, twx_result.SNRr + SNRi) follows the true SNR,
    (ii) the cross-correlation estimate peak^2 / bruit saturates at high SNR (the code's own correlation sidelobes become the "noise"),
    (iii) the squared-spectrum estimate valmax_square^2 / noise_square rises with the signal as well.  All three from ONE pass of the
    device path, every number equal to the oracle's (previous tests); here the behaviour the README's table shows."""
    nchips = 100000
    chips = prn.lfsr_chips(17, 9, nchips)
    n = 2 * nchips
    amps = [80, 253, 800, 2530, 8000, 25300]                           # power steps of 10 dB, -17 dB ... +33 dB; sigma = 400 per component throughout
    # (below about -20 dB the carrier search on the squared signal no longer finds the carrier in a 40-ms window: the README's first problem)
    # The carrier is handed over (processing(d,df)) for the two correlation-based estimates: the coarse estimator's own grid — df = freq(idx)/2 on
    # the linspace axis, godual_ranging.m:15,73 — sits a quarter of a bin off the transform's bins, so that its best answer leaves 6 Hz over a
    # 40-ms window, 1.5 rad of phase drift, which becomes the wipe-off's own noise floor (SNR estimate 0.58 whatever the signal: the oracle says
    # the same).  The squared-spectrum estimate comes from a second call with the band.
    df_true = 900.0
    ps = [synth.SynthParams(delay_q8=4321 * 256, fstep=synth.fstep_for_df(df_true, FS), phi0=5, amp=a, noise_gain=synth.noise_gain_for_sigma(400.0), seed=11 + i)
          for i, a in enumerate(amps)]
    raw = np.concatenate([synth.synth_channel(n, chips, 2, p) for p in ps]).reshape(-1)
    band = band_godual(FS, n)
    with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as cor:
        cor.set_snr_estimators(10001, 10001)
        got = cor.process(raw, n_channels=1, channel=0, df=df_true)
        ex = cor.snr_estimators(len(amps))
        got_b = cor.process(raw, n_channels=1, channel=0, band=band)
        ex_b = cor.snr_estimators(len(amps))
    true = np.array([a * a / (2 * 400.0 ** 2) for a in amps])
    wipe = np.array([g.SNRr + g.SNRi for g in got])
    xc = np.array([abs(g.xval) ** 2 / e["bruit"] for g, e in zip(got, ex)])
    sq = np.array([e["valmax_square"] ** 2 / e["noise_square"] for e in ex_b])
    assert all(g.indice == 3 * 4321 for g in got) and all(g.indice == 3 * 4321 for g in got_b)
    print("true SNR", true, "\nwipe-off", wipe, "\nxcorr peak^2/bruit", xc, "\nsquared spectrum", sq)
    # (i) the wipe-off estimate is the SNR — a decade of signal power is a decade of estimate — over the range of a TWSTFT link (README: -20 ...
    # -5 dB) and up to +3 dB; above, the ripple the band-limited x3 interpolation leaves on the held chips becomes x.*code's own variance and
    # the estimate levels off at +10 dB for this two-samples-per-chip signal (the fp64 oracle: 6.46, 9.53, 9.99 for +13, +23, +33 dB)
    lo = true <= 3.0
    assert np.all((wipe[lo] / true[lo] > 0.6) & (wipe[lo] / true[lo] < 1.2)), wipe / true
    assert np.all(np.abs(np.diff(np.log10(wipe[lo])) - np.diff(np.log10(true[lo]))) < 0.1), (wipe, true)
    assert np.all(np.diff(wipe) > 0)
    # (ii) the correlation estimate grows while noise dominates the sidelobes and then stops: the last decade of signal power buys < 2x
    assert xc[2] / xc[0] > 20 and xc[-1] / xc[-2] < 2.0, xc
    # (iii) the squared-spectrum estimate keeps rising with the signal
    assert np.all(np.diff(np.log(sq)) > 0), sq
