"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden fixtures.

Gates (BASELINE.json north_star): integer-sample peak lag bit-exact; peak magnitude within 1e-6
relative (fp32 device path vs fp64 oracle); correction / df / SNR compared at tolerances stated
per assertion.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from amaranth_twstft_amd import prn, synth
from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd.correlator import ALL_CHANNELS, Correlator, band_numpy, band_godual, lfsr_chips_device
from oracle import twstft_oracle as orc
from tests.helpers import load_golden, capture_from_desc, chips_for

pytestmark = pytest.mark.gpu
FS = 5e6
MAG_TOL = 1e-6          # north_star: peak magnitude within 1e-6 relative (fp32)


def _check(g, o, corr_tol=2e-4, snr_tol=2e-4):
    assert g.indice == o["indice"]                                            # bit-exact integer lag
    assert abs(abs(g.xval) - abs(o["xval"])) <= MAG_TOL * abs(o["xval"])
    assert abs(g.xval - o["xval"]) <= 2 * MAG_TOL * abs(o["xval"])
    assert abs(abs(g.xvalm1) - abs(o["xvalm1"])) <= 2 * MAG_TOL * abs(o["xval"])
    assert abs(abs(g.xvalp1) - abs(o["xvalp1"])) <= 2 * MAG_TOL * abs(o["xval"])
    assert abs(g.correction - o["correction"]) <= corr_tol                    # samples of the 3x grid (<= 13 ps)
    assert abs(g.df - o["df"]) <= 1e-9
    for k in ("SNRr", "SNRi", "puissancecode"):
        assert abs(getattr(g, k) - o[k]) <= snr_tol * max(abs(o["SNRr"]), abs(o["SNRi"]), abs(o[k])) + 1e-30, k
    assert abs(g.puissance - o["puissance"]) <= 1e-6 * abs(o["puissance"])
    # var(yincode) = mean|yint|^2 - |mean|^2 is a difference: the fp32 peak samples (1e-7) enter through |mean|^2
    assert abs(g.puissancenoise - o["puissancenoise"]) <= 1e-6 * abs(o["puissancenoise"]) + 5e-7 * abs(o["puissancecode"])


def _capture(bitlen, taps, nchips, nwin, seed, df=(1780.75, 0.0), amp=(300, 3000), sigma=(500.0, 100.0)):
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    chans = [synth.SynthParams(delay_q8=(n // 3 + 1157) * 256, fstep=synth.fstep_for_df(df[0], FS), phi0=1 << 29, amp=amp[0],
                               noise_gain=synth.noise_gain_for_sigma(sigma[0]), seed=seed, stream=0),
             synth.SynthParams(delay_q8=(n // 5) * 256, fstep=synth.fstep_for_df(df[1], FS), phi0=0, amp=amp[1],
                               noise_gain=synth.noise_gain_for_sigma(sigma[1]), seed=seed, stream=1)]
    return chips, synth.synth_capture(n * nwin, chips, 2, chans)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("bitlen,taps,nchips", [(13, 27, 5000), (14, 43, 10000), (15, 3, 25000), (17, 9, 100000),
                                                 (18, 39, 250000), (19, 39, 500000),
                                                 # plug-in lengths: N = 5000, 25000, 4000, 80000, 81000, 14000, 6000, 12000, 18000, 2^16, 2^19, 2^20
                                                 (13, 27, 2500), (15, 3, 12500), (12, 83, 2000), (16, 45, 40000), (16, 45, 40500), (13, 27, 7000),
                                                 (12, 83, 3000), (13, 27, 6000), (14, 43, 9000), (16, 45, 32768), (18, 39, 262144), (19, 39, 524288)])
def test_fft_and_code_spectrum(bitlen, taps, nchips, precision):
    """The N-point transform of the context (twx_fft_forward: column pass + ROW_STORE row pass) and its code spectrum against numpy,
    every output, for every window length the library or its plug-ins hold, in both precisions."""
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    with Correlator(chips, fs=FS, precision=precision) as cor:
        rng = np.random.default_rng(n)
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        ref = np.fft.fft(x)
        assert np.abs(cor.fft(x) - ref).max() <= (1e-6 if precision == "f32" else 1e-13) * np.abs(ref).max()
        cref = orc.make_fcode(orc.make_code(chips, 2))
        assert np.abs(cor.code_spectrum() - cref).max() <= (2e-7 if precision == "f32" else 1e-13) * np.abs(cref).max()


@pytest.mark.parametrize("bitlen,taps,nchips,nwin", [(13, 27, 5000, 5), (14, 57, 10000, 4), (15, 17, 25000, 3),
                                                      (17, 15, 100000, 3), (18, 63, 250000, 2), (19, 63, 500000, 1)])
def test_processing_vs_oracle_two_channels(bitlen, taps, nchips, nwin):
    chips, raw = _capture(bitlen, taps, nchips, nwin, seed=nchips)
    n = 2 * nchips
    band = band_numpy(FS, n)
    with Correlator(chips, fs=FS, Nint=1) as cor:
        got = cor.ranging(raw, n_channels=2, channels=(0, 1), band=band)
    ref = orc.ranging(raw, chips, fs=FS, Nint=1, n_channels=2, band="numpy")
    for c in (0, 1):
        assert len(got[c]) == len(ref[c]) == nwin
        for g, o in zip(got[c], ref[c]):
            _check(g, o)
            assert g.df_index >= 0


def test_df_supplied_and_full_map():
    chips, raw = _capture(14, 43, 10000, 2, seed=3)
    n = 20000
    with Correlator(chips, fs=FS, Nint=1) as cor:
        got = cor.process(raw, n_channels=2, channel=0, df=[1781.0, 1779.5])
        z = cor.xcorr_map(raw[:n], 1781.0, n_channels=2, channel=0)
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    temps = np.arange(n) / FS
    for w, df in enumerate((1781.0, 1779.5)):
        d = orc.deinterleave(raw[w * n:(w + 1) * n], 2, 0)
        d = d - d.mean()
        o = orc.processing(d, None, None, temps, fcode, code, Nint=1, fs=FS, df=df)
        _check(got[w], o)
        assert got[w].df_index == -1
        if w == 0:
            zr = orc.xcorr_interp(np.fft.fft(d * np.exp(-2j * np.pi * df * temps)), fcode, 1)
            assert np.abs(z - zr).max() <= 1e-6 * np.abs(zr).max()
            assert int(np.abs(z).argmax()) == o["indice"]
            # the seven peak samples and the 3/5/7-point parabola fits of 221207 godual_ranging.m:73-78
            zw = zr[(o["indice"] + np.arange(-3, 4)) % len(zr)]
            assert np.abs(got[w].zwin - zw).max() <= 2 * MAG_TOL * abs(o["xval"])
            for h in (1, 2, 3):
                assert abs(got[w].correction_polyfit(h) - orc.peak_refine_polyfit(zr, o["indice"], h)) < 2e-4
            assert abs(got[w].correction_polyfit(1) - got[w].correction) < 1e-9       # 3 points: the same parabola as :33


@pytest.mark.parametrize("Nint", [0, 2])
def test_other_interpolation_factors(Nint):
    chips, raw = _capture(14, 43, 10000, 2, seed=9)
    n = 20000
    with Correlator(chips, fs=FS, Nint=Nint) as cor:
        got = cor.process(raw, n_channels=2, channel=0, band=band_numpy(FS, n))
    ref = orc.ranging(raw, chips, fs=FS, Nint=Nint, n_channels=2, channels=(0,), band="numpy")[0]
    for g, o in zip(got, ref):
        _check(g, o)


def test_octave_variance_convention_and_remote_band():
    chips, raw = _capture(14, 43, 10000, 1, seed=21, df=(50000.0, 0.0), amp=(800, 3000))
    n = 20000
    with Correlator(chips, fs=FS, Nint=1, var_ddof=1) as cor:
        g = cor.processing(raw, band_godual(FS, n, remote=1, OP=0), n_channels=2, channel=0)
    o = orc.ranging(raw, chips, fs=FS, Nint=1, n_channels=2, channels=(0,), band="godual", remote=1, OP=0, ddof=1)[0][0]
    _check(g, o)
    assert abs(g.df - 50000.0) < 300.0


def test_single_channel_layout_and_short_final_window():
    chips = chips_for(14, 43, 10000)
    n = 20000
    p = synth.SynthParams(delay_q8=4321 * 256, fstep=synth.fstep_for_df(-700.0, FS), phi0=5, amp=250,
                          noise_gain=synth.noise_gain_for_sigma(400.0), seed=77)
    raw = synth.synth_channel(2 * n + 1234, chips, 2, p)           # 2 full windows + a short one
    with Correlator(chips, fs=FS) as cor:
        got = cor.process(raw, n_channels=1, channel=0, band=band_numpy(FS, n))
        assert cor.process(raw[:100], n_channels=1, channel=0, band=band_numpy(FS, n)) == []   # empty: no full window
    assert len(got) == 2                                             # godual_ranging.m:81,102
    ref = orc.ranging(raw[:2 * n], chips, fs=FS, Nint=1, n_channels=1, channels=(0,), band="numpy")[0]
    for g, o in zip(got, ref):
        _check(g, o)
        assert g.indice == 3 * 4321


@pytest.mark.parametrize("case", ["c5k", "c10k", "c25k", "c100k", "c250k", "c500k"])
def test_golden_221207_rows(case):
    """Device path reproduces the reference's own printed rows (tests/golden/ref221207_ranging.json)."""
    g = load_golden("ref221207_ranging.json")
    c = next(x for x in g["cases"] if x["name"] == case)
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    n = 2 * len(chips)
    fs = c["fs"]
    with Correlator(chips, fs=fs, Nint=c["Nint"], snr_rot=-2) as cor1, Correlator(chips, fs=fs, Nint=c["Nint"], snr_rot=0) as cor2:
        r1 = cor1.process(raw, n_channels=2, channel=0, band=band_numpy(fs, n))
        r2 = cor2.process(raw, n_channels=2, channel=1, df=0.0)     # ch2 is not mixed in that script (:66)
    for a, b, line in zip(r1, r2, c["rows"]):
        ref = [float(v) for v in line.split("\t")[1:]]
        mine = ((a.indice - b.indice + a.correction - b.correction) / fs / 3, a.df, 10 * np.log10(a.puissance),
                10 * np.log10(a.SNRi + a.SNRr), 10 * np.log10(b.SNRi + b.SNRr))
        assert abs(mine[0] - ref[0]) <= 1.0e-12 + 2e-4 / fs / 3      # printed to 1e-12 s; fp32 correction tolerance
        assert abs(mine[1] - ref[1]) <= 0.05 + 1e-9
        for x, y in zip(mine[2:], ref[2:]):
            assert abs(x - y) <= 0.051


def test_golden_220830_op_rows():
    """Device path with the zero-mean 0/1 replica reproduces the rows printed by the reference's own
    experiments/220830_OP/godual_ranging_OP.py (tests/golden/ref220830_op_ranging.json): lag bit-exact, correction,
    and the COMPLEX peak sample (magnitude 1e-6, phase through the complex difference)."""
    g = load_golden("ref220830_op_ranging.json")
    c = g["cases"][0]
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    fs, n = g["fs"], 2 * len(chips)
    with Correlator(chips, fs=fs, Nint=g["Nint"], code_levels="unipolar", code_zero_mean=True) as cor:
        got = cor.process(raw, n_channels=2, channel=0, band=band_numpy(fs, n))
    assert len(got) == c["nwin"] == len(c["rows"])
    for a, ref in zip(got, c["rows"]):
        z = complex(*ref["xval"])
        assert a.indice == ref["indice"]
        assert abs(a.correction - ref["correction"]) <= 2e-4
        assert abs(abs(a.xval) - abs(z)) <= MAG_TOL * abs(z)
        assert abs(a.xval - z) <= 5e-6 * abs(z)


def test_full_size_window_vs_golden_and_oracle():
    """N = 5 000 000 (BASELINE.json configs[1]): lag equals the reference's (221219 golden, C2) and
    everything else matches the oracle run on the same input."""
    g = load_golden("ref221219_processing.json")
    c = next(x for x in g["cases"] if x["name"] == "n5M_C2")
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    n = 2 * len(chips)
    band = band_numpy(FS, n)
    with Correlator(chips, fs=FS, Nint=1) as cor:
        assert (cor.info.n1, cor.info.n2) == (625, 8000)
        got = cor.process(raw, n_channels=1, channel=0, band=band)[0]
    assert got.indice == c["ref"]["indice"] == 3 * 1311765           # reference's own result
    code = orc.make_code(chips, 2)
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    freq = orc.freq_axis(FS, n)
    o = orc.processing(d, np.arange(band[0], band[1] + 1), freq, np.arange(n) / FS, orc.make_fcode(code), code, Nint=1, fs=FS)
    _check(got, o)


def test_batched_device_api_linearity_and_determinism():
    """Size-independent properties at full size on device-resident data: every window of a batch
    peaks at 3*delay, results are independent of batch size, and repeated runs are bit-identical."""
    import torch
    lib = L.load()
    nchips, n = 2500000, 5000000
    chips = chips_for(22, 3, nchips)
    dev = torch.device("cuda", 0)
    chips_dev = torch.from_numpy(chips).to(dev)
    nwin = 6
    iq = torch.empty((nwin, n, 2), dtype=torch.int16, device=dev)
    delays = [1311765 - 7 * p for p in range(nwin)]
    for p in range(nwin):
        params = np.array([delays[p] * 256, synth.fstep_for_df(1780.75, FS), 12345 * p, 200, synth.noise_gain_for_sigma(400.0),
                           1000 + p, 0, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(iq[p].data_ptr(), n, 0, chips_dev.data_ptr(), nchips, 2, 1,
                                          params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    # device generator == numpy generator (bit-exact) on a slice
    ref = synth.synth_channel(4096, chips, 2, synth.SynthParams(delay_q8=delays[1] * 256, fstep=synth.fstep_for_df(1780.75, FS),
                                                              phi0=12345, amp=200, noise_gain=synth.noise_gain_for_sigma(400.0),
                                                              seed=1001, stream=0), n0=0)
    assert np.array_equal(iq[1, :4096].cpu().numpy(), ref)
    band = L.twx_band(*band_godual(FS, n))
    outs = []
    # 1: 625 rows per launch (one per workgroup of the row pass); 3: 1875 rows on 1280 row-walking workgroups (one or two rows
    # each); 4: 2500; 0: the default batch
    for batch in (1, 3, 4, 0):
        with Correlator(chips, fs=FS, Nint=1, max_batch=batch) as cor:
            res = torch.zeros((nwin, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
            for _ in range(2):
                L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
                L.check(lib.twx_synchronize(cor._h), cor._h)
                outs.append(res.cpu().numpy().tobytes())
    assert all(o == outs[0] for o in outs)                           # batch-size independent, run-to-run identical
    arr = (L.twx_result * nwin).from_buffer_copy(outs[0])
    for p in range(nwin):
        assert arr[p].indice0 == 3 * delays[p]
        assert abs(arr[p].df - 1780.75) < 0.51                        # 0.5 Hz bins (godual_ranging.m:14)


def test_lfsr_device_generator():
    for bitlen, taps, n in [(13, 27, 5000), (17, 9, 100000), (22, 57, 300001)]:
        assert np.array_equal(lfsr_chips_device(bitlen, taps, n), prn.lfsr_chips(bitlen, taps, n))
    g = load_golden("prn_codes.json")
    e = next(x for x in g["files"] if x["name"] == "noiselen2500000_bitlen22_taps03.bin.gz")
    assert hashlib.sha256(lfsr_chips_device(22, 3, e["len"]).tobytes()).hexdigest() == e["sha256"]


def test_context_from_lfsr_parameters_equals_context_from_chips():
    chips, raw = _capture(14, 43, 10000, 1, seed=5)
    with Correlator(chips, fs=FS) as a, Correlator(lfsr=(14, 43, 10000), fs=FS) as b:
        ra = a.process(raw, n_channels=2, channel=0, band=band_numpy(FS, 20000))[0]
        rb = b.process(raw, n_channels=2, channel=0, band=band_numpy(FS, 20000))[0]
    assert ra.indice == rb.indice and ra.xval == rb.xval and ra.SNRr == rb.SNRr and np.array_equal(ra.zwin, rb.zwin)


def test_unsupported_length_is_a_clean_error():
    """A window length with a prime factor other than 2, 3, 5 (9998 = 2 x 4999) has no N1 x N2 plan: the C entry point
    answers TWX_E_SIZE with a message that says what to do, the Python wrapper (which would otherwise build a plan
    plug-in) says why none can exist."""
    chips = chips_for(13, 27, 5000)[:4999]
    lib = L.load()
    cfg = L.twx_config()
    cfg.fs, cfg.sps, cfg.nint, cfg.n_chips = FS, 2, 1, chips.size
    cfg.chips = chips.ctypes.data_as(C.POINTER(C.c_uint8))
    h = C.c_void_p()
    assert lib.twx_create(C.byref(cfg), C.byref(h)) == -2
    assert b"no plan pair" in lib.twx_last_error(None) and b"amaranth_twstft_amd.plans" in lib.twx_last_error(None)
    with pytest.raises(ValueError) as e:
        Correlator(chips, fs=FS)
    assert "2^a 3^b 5^c" in str(e.value)


def test_fp64_context_tighter_than_fp32():
    """TWX_F64 instantiation of the same kernels (BASELINE.json configs[4]: fp64 vs fp32 tolerance)."""
    chips, raw = _capture(17, 9, 100000, 1, seed=31)
    n = 200000
    band = band_numpy(FS, n)
    with Correlator(chips, fs=FS, Nint=1, precision="f64") as c64, Correlator(chips, fs=FS, Nint=1) as c32:
        g64 = c64.process(raw, n_channels=2, channel=0, band=band)[0]
        g32 = c32.process(raw, n_channels=2, channel=0, band=band)[0]
    o = orc.ranging(raw, chips, fs=FS, Nint=1, n_channels=2, channels=(0,), band="numpy")[0][0]
    assert g64.indice == g32.indice == o["indice"]
    assert abs(g64.xval - o["xval"]) <= 1e-10 * abs(o["xval"])       # both sides are fp64 FFTs of 2e5 points
    assert abs(g64.correction - o["correction"]) <= 1e-9
    assert abs(g64.SNRr - o["SNRr"]) <= 1e-9 * o["SNRr"]
    assert abs(abs(g32.xval) - abs(g64.xval)) <= MAG_TOL * abs(g64.xval)       # fp32 vs fp64 peak magnitude


def test_hamming_windowed_code_spectrum():
    """fcode × Hamming (processing/CPP/main.cpp:717-719); oracle restatement is unpinned."""
    chips, raw = _capture(14, 43, 10000, 1, seed=41)
    n = 20000
    with Correlator(chips, fs=FS, Nint=1, window="hamming") as cor:
        cs = cor.code_spectrum()
        g = cor.process(raw, n_channels=2, channel=1, df=0.0)[0]
    code = orc.make_code(chips, 2)
    fh = orc.make_fcode(code, "hamming")
    assert np.abs(cs - fh).max() <= 3e-7 * np.abs(fh).max()
    d = orc.deinterleave(raw, 2, 1)
    d = d - d.mean()
    z = orc.xcorr_interp(np.fft.fft(d), fh, 1)
    ind, corr, xval, _, _ = orc.peak_refine(z)
    assert g.indice == ind and abs(abs(g.xval) - abs(xval)) <= MAG_TOL * abs(xval) and abs(g.correction - corr) < 2e-4
    # the wipe-off statistics: the C++ twin takes them from yint = ifft(zero-padded FFT(y)) — no window in it (main.cpp:319-332) —
    # rotated to the peak of the WINDOWED map; orc.processing does the same (its SNR never sees fcode)
    o = orc.processing(d, None, None, np.arange(n) / FS, fh, code, Nint=1, fs=FS, df=0.0)
    _check(g, o)


def test_claudio_convention():
    """Reversed-conjugate convention of acquisition/claudio_aligned_code_ranging_separate.m:49-102."""
    chips, raw = _capture(17, 9, 100000, 2, seed=51, df=(300.0, 0.0))
    n = 200000
    with Correlator(chips, fs=FS, Nint=1, convention="claudio", var_ddof=1) as cor:
        got = cor.process(raw, n_channels=2, channel=0, df=300.0)
    code = orc.make_code(chips, 2)
    temps = np.arange(n) / FS
    fc = orc.make_fcode(code, "claudio")
    for w, g in enumerate(got):
        d = orc.deinterleave(raw[w * n:(w + 1) * n], 2, 0)
        d = d - d.mean()
        o = orc.processing_claudio(d, 300.0, temps, fc, code, Nint=1, ddof=1)
        _check(g, o)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("bitlen,taps,nchips,delay", [(14, 43, 10000, 6543), (13, 27, 5000, 399), (16, 45, 32768, 255 + 256 * 9), (17, 9, 100000, 4000 * 7 - 1)])
def test_caf_integer_bins_vs_oracle(bitlen, taps, nchips, delay, precision):
    """Delay x Doppler surface on the integer-bin grid vs the oracle's
    shift-and-correlate restatement of rxcomplex.cpp:534-563 (unpinned) — in both precisions, over the row-pass forms of the
    surface kernels (N2 = 400 and 256: k_row_caf; 4000: k_rowd_caf), the peak on the LAST element of a row of the two-pass layout."""
    n = 2 * nchips
    chips = chips_for(bitlen, taps, nchips)
    df_bins = 7                                               # true offset = 7 bins = 7*fs/N
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(df_bins * FS / n, FS), phi0=77, amp=400,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=8)
    raw = synth.synth_channel(n, chips, 2, p)
    with Correlator(chips, fs=FS, Nint=0, precision=precision) as cor:
        assert delay % int(cor.info.n2) == int(cor.info.n2) - 1 or nchips == 10000
        pk, lag = cor.caf_bins(raw, -40, 40)
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    ks, pk_o, lag_o = orc.caf_bins_shift(d, orc.make_fcode(orc.make_code(chips, 2)), -40, 40)
    assert np.array_equal(lag, lag_o)                          # every bin's arg-max, bit-exact
    assert np.abs(pk - pk_o).max() <= (MAG_TOL if precision == "f32" else 1e-12) * pk_o.max()
    best = int(np.argmax(pk))
    assert ks[best] == df_bins and lag[best] == delay


def test_caf_row_pass_forms_agree_on_random_windows(monkeypatch):
    """k_rowd_caf (DIF/DIT, several bins per workgroup, rotated block-thread addressing) against k_row_caf (Stockham) on
    seeded windows from strong signal to pure noise, bins on both sides of zero and far enough out that the k2 rotation
    carries (|kappa| > N1): every bin's lag identical, peaks within 1e-6."""
    nchips, n = 100000, 200000                                   # N1 = 50, N2 = 4000
    chips = chips_for(17, 9, nchips)
    rng = np.random.default_rng(4242)
    with Correlator(chips, fs=FS, Nint=0) as cor:
        assert (cor.info.n1, cor.info.n2) == (50, 4000)
        for amp in (600, 60, 0):
            p = synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256, fstep=synth.fstep_for_df(float(rng.uniform(-3000, 3000)), FS),
                                  phi0=int(rng.integers(0, 1 << 30)), amp=amp, noise_gain=synth.noise_gain_for_sigma(500.0),
                                  seed=int(rng.integers(1, 1 << 20)))
            raw = synth.synth_channel(n, chips, 2, p)
            for lo, hi in ((-130, 75), (9_990, 10_060), (-100_003, -99_950)):
                monkeypatch.delenv("TWX_CAF_STOCKHAM", raising=False)
                pk_d, lag_d = cor.caf_bins(raw, lo, hi)
                monkeypatch.setenv("TWX_CAF_STOCKHAM", "1")
                pk_s, lag_s = cor.caf_bins(raw, lo, hi)
                assert np.array_equal(lag_d, lag_s), (amp, lo, hi)
                assert np.abs(pk_d - pk_s).max() <= MAG_TOL * pk_s.max()
    monkeypatch.delenv("TWX_CAF_STOCKHAM", raising=False)


def test_caf_arbitrary_frequencies_and_acquire():
    nchips, n = 10000, 20000
    chips = chips_for(14, 57, nchips)
    p = synth.SynthParams(delay_q8=1500 * 256, fstep=synth.fstep_for_df(1337.0, FS), phi0=5, amp=500,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=18)
    raw = synth.synth_channel(n, chips, 2, p)
    freqs = np.array([1000.0, 1250.0, 1337.0, 1500.0])
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    fcode = orc.make_fcode(orc.make_code(chips, 2))
    with Correlator(chips, fs=FS, Nint=0) as cor:
        res = cor.caf_freqs(raw, freqs)
        for f, r in zip(freqs, res):
            y = d * np.exp(-2j * np.pi * f * np.arange(n) / FS)
            m = np.abs(np.fft.ifft(np.fft.fft(y) * fcode))
            assert r.indice == int(m.argmax())
            assert abs(abs(r.xval) - m.max()) <= MAG_TOL * m.max()
        fc, pk, lag = cor.acquire(raw, fc_init=1200.0, frange=512.0, fstep=128.0)
    assert lag == 1500 and abs(fc - 1337.0) <= 1.0


def test_caf_full_size_peak_location():
    """Full-size window (5e6 samples): the CAF peak over +-30 bins sits at the generator's offset and delay."""
    nchips, n = 2500000, 5000000
    chips = chips_for(22, 3, nchips)
    p = synth.SynthParams(delay_q8=1311765 * 256, fstep=synth.fstep_for_df(12.0, FS), phi0=0, amp=200,
                          noise_gain=synth.noise_gain_for_sigma(400.0), seed=7)
    import ctypes as C2
    lib = L.load()
    import torch
    dev = torch.device("cuda", 0)
    iq = torch.empty((n, 2), dtype=torch.int16, device=dev)
    params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
    cd = torch.from_numpy(chips).to(dev)
    L.check(lib.twx_synth_capture_dev(iq.data_ptr(), n, 0, cd.data_ptr(), nchips, 2, 1, params.ctypes.data_as(C2.c_void_p), None))
    torch.cuda.synchronize()
    raw = iq.cpu().numpy()
    with Correlator(chips, fs=FS, Nint=0) as cor:
        pk, lag = cor.caf_bins(raw, -30, 30)
    best = int(np.argmax(pk))
    assert best - 30 == 12 and lag[best] == 1311765
    assert pk[best] > 3 * np.median(pk)


def test_sliding_dot_short_code():
    """Direct path (a12): ±nlag sliding dot products per 40-ms code period vs the oracle restatement
    of rxcomplex.cpp:605,989-999 (unpinned) and vs the FFT path on the same data."""
    from amaranth_twstft_amd import tracking
    nchips, sps = 10000, 2
    nobs, ncodes, nlag = nchips * sps, 5, 28
    chips = chips_for(14, 43, nchips)
    delay = 11
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(250.0, FS), phi0=1 << 28, amp=600,
                          noise_gain=synth.noise_gain_for_sigma(200.0), seed=61)
    raw = synth.synth_channel(nobs * ncodes + 100, chips, sps, p)
    code = orc.make_code(chips, sps)
    ff = synth.df_of_fstep(p.fstep, FS) / FS
    got = tracking.sliding_dot(raw, code, nobs, ncodes, nlag, pt=0, ff=ff, phi=0.0, scale=np.sqrt(2.0) / 32768.0)
    x = orc.deinterleave(raw, 1, 0)
    for pp in range(ncodes):
        i = np.arange(pp * nobs, (pp + 1) * nobs)
        y = np.sqrt(2.0) / 32768.0 * x[i] * np.exp(-2j * np.pi * ff * i)
        ref = orc.sliding_dot(y, code, nlag)
        assert np.abs(got[pp] - ref).max() <= 2e-6 * np.abs(ref).max()
        assert int(np.abs(got[pp]).argmax()) - nlag == delay
    cor, phi = tracking.get_cor_and_phi(got)
    lag, hrc = tracking.hrc_delay(cor, nlag)
    assert (lag == delay).all() and np.all(np.abs(hrc - delay) < 0.5)


@pytest.mark.parametrize("nobs,ncodes,nlag,nch,ch,pt", [
    (20000, 5, 28, 1, 0, 0),          # one chunk per workgroup, a ragged last pass
    (400000, 3, 28, 1, 0, 3),         # sdr.param size: chunks of several LDS pieces, odd start
    (40001, 2, 31, 1, 0, 1),          # odd period: the replica segment wraps at an odd index
    (17000, 4, 16, 2, 1, 5),          # second channel of a two-channel capture
    (9000, 7, 8, 1, 0, 0),
    (5000, 3, 4, 1, 0, 2),
    (700, 2, 28, 1, 0, 0),            # a period shorter than one pass of the workgroup
    (10, 3, 28, 1, 0, 0),             # a period shorter than the lag window
    (1, 2, 4, 1, 0, 0),
    # the streaming form of narrow windows (nlag <= 8, one channel, nobs a multiple of 8: eight samples per lane, 16-byte loads)
    (400000, 24, 8, 1, 0, 3),         # sdr.param size at the program's default +-8 lags (rxcomplex.cpp:295), unaligned start
    (40000, 30, 4, 1, 0, 1),          # 4-ms codes
    (8200, 2, 8, 1, 0, 5),            # a chunk one group longer than the 8192-sample LDS piece
    (8, 3, 4, 1, 0, 0),               # a period of one group
    (40000, 5, 8, 2, 1, 0),           # two channels
    (9004, 3, 8, 1, 0, 0),            # period not a multiple of 8: the general form
    # wide windows: eight samples per lane where the period is a multiple of 8 and the capture has one channel (the shapes above
    # with nobs % 8 == 0 and nlag > 8 take it too), four otherwise
    (6152, 2, 31, 1, 0, 1),           # three passes of 2048 and one group
    (40008, 3, 16, 1, 0, 7),
    (16, 2, 28, 1, 0, 0),             # a period of two groups, shorter than the lag window
    (20004, 5, 28, 1, 0, 0),          # not a multiple of 8: four samples per lane
    # a channel of a two-channel capture (the reference's file format) in the eight-samples-per-lane forms: whole frames loaded,
    # the channel's words picked out ((40000, 5, 8, 2, 1) and (17000, 4, 16, 2, 1) above take them too)
    (40000, 3, 4, 2, 0, 3),
    (16392, 2, 28, 2, 0, 0),          # one group more than the LDS piece
    (16392, 2, 28, 2, 1, 1),
])
def test_sliding_dot_shapes(nobs, ncodes, nlag, nch, ch, pt):
    """k_sliding_dot over its lag-count instantiations (4, 8, 16, 28, 31), chunk / piece / pass boundaries, odd periods (the wrap
    of the replica segment), a channel of two and a start offset, against the definition in fp64 (out[p][l] = scale/nobs *
    sum_i x[pt + p nobs + i] e^{-2 pi j (ff (p nobs + i) + phi)} w[(i - lag) mod nobs], = downconv_trk + the replica dgemm of
    rxcomplex.cpp:605,1051-1061)."""
    from amaranth_twstft_amd import tracking
    rng = np.random.default_rng(nobs + 7 * nlag)
    w = rng.choice([-1.0, 1.0], nobs).astype(np.float32)
    raw = np.clip(rng.normal(0, 3000, (pt + nobs * ncodes + 8, 2 * nch)), -32768, 32767).astype(np.int16)
    ff, phi, scale = 3.1e-5, 0.37, 1.0 / 32768.0
    got = tracking.sliding_dot(raw, w, nobs, ncodes, nlag, pt=pt, ff=ff, phi=phi, scale=scale, n_channels=nch, channel=ch)
    x = raw[:, 2 * ch].astype(np.float64) + 1j * raw[:, 2 * ch + 1]
    for p in range(ncodes):
        i = np.arange(p * nobs, (p + 1) * nobs)
        y = scale * x[pt + i] * np.exp(-2j * np.pi * (ff * i + phi))
        ref = orc.sliding_dot(y, w.astype(np.float64), nlag)
        assert np.abs(got[p] - ref).max() <= 3e-6 * np.abs(ref).max() + 1e-9, (p, np.abs(got[p] - ref).max(), np.abs(ref).max())


def test_fir_decimating_front_end():
    """70 Msps → 5 Msps front end (configs[4]); oracle = fp64 direct convolution (unpinned)."""
    from amaranth_twstft_amd import frontend
    fs_in, dec = 70e6, 14
    taps = frontend.lowpass_taps(fs_in, 2.1e6, 0.4e6)
    assert taps.size % 2 == 1 and 400 < taps.size < 1024
    assert np.allclose(taps, orc.fir_lowpass_hamming(fs_in, 2.1e6, 0.4e6), atol=1e-7)
    rng = np.random.default_rng(5)
    n_in = 200000
    raw = np.clip(rng.normal(0, 3000, (n_in, 2)), -32768, 32767).astype(np.int16)
    y = frontend.fir_decimate(raw, taps, dec, out="f32")
    x = raw[:, 0].astype(np.float64) + 1j * raw[:, 1]
    ref = orc.fir_decimate(x, taps.astype(np.float64), dec)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-3
    y16 = frontend.fir_decimate(raw, taps, dec, out="int16")
    assert np.abs(y16[:, 0] - np.rint(ref.real)).max() <= 1 and np.abs(y16[:, 1] - np.rint(ref.imag)).max() <= 1


@pytest.mark.parametrize("ntaps,dec,nch,ch", [
    (421, 14, 1, 0),      # configs[4]: 31 taps per phase, 9 step groups, two live steps in the last group
    (577, 14, 1, 0),      # the longer design of the SURVEY (42 taps per phase)
    (64, 16, 1, 0),       # 4 taps per phase: fewer step groups than the smallest unrolled kernel
    (33, 3, 1, 0),        # 11 taps per phase, three live steps in the last group
    (100, 5, 1, 0),       # 20 taps per phase, a full last group
    (97, 4, 1, 0),        # 25 taps per phase, one live step in the last group
    (330, 5, 1, 0),       # 66 taps per phase: the generic (not unrolled) kernel
    (421, 14, 2, 1),      # second channel of a two-channel capture: the 4-byte staging path
    (57, 7, 2, 0),
    (171, 3, 1, 0),       # 57 taps per phase: the longest the eight-output form takes (16 step groups), odd phase count (2 + 1)
    (232, 4, 1, 0),       # 58 taps per phase: back to the four-output form
    (31, 1, 1, 0),        # no decimation: one phase, four-output form
])
def test_fir_decimator_shapes(ntaps, dec, nch, ch):
    """Both forms of the kernel (k_fir_poly8: eight outputs per thread, phases split between the halves of a workgroup; k_fir_poly:
    four outputs), every step-group count (unrolled 4..16, the generic loop), every count of live steps in the last group, both
    staging paths (16-byte loads of one aligned channel, 4-byte loads of a channel of two) and a ragged last workgroup, against
    the fp64 direct sum (orc.fir_decimate, the oracle's definition; unpinned by nature: the reference has no such filter)."""
    from amaranth_twstft_amd import frontend
    rng = np.random.default_rng(1000 * ntaps + dec)
    taps = (rng.normal(0, 1, ntaps) * np.hamming(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    n_in = 14 * 2048 + 3 * ntaps + 11                                   # several workgroups and a ragged tail
    raw = np.clip(rng.normal(0, 4000, (n_in, 2 * nch)), -32768, 32767).astype(np.int16)
    y = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="f32")
    x = raw[:, 2 * ch].astype(np.float64) + 1j * raw[:, 2 * ch + 1]
    ref = orc.fir_decimate(x, taps.astype(np.float64), dec)
    assert y.shape == ref.shape and y.shape[0] == (n_in - ntaps) // dec + 1
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-3
    y16 = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="int16")
    assert np.abs(y16[:, 0] - np.clip(np.rint(ref.real), -32768, 32767)).max() <= 1
    assert np.abs(y16[:, 1] - np.clip(np.rint(ref.imag), -32768, 32767)).max() <= 1
    # the shortest inputs: one and two outputs
    for extra in (0, dec):
        ys = frontend.fir_decimate(raw[: ntaps + extra], taps, dec, n_channels=nch, channel=ch, out="f32")
        assert ys.shape[0] == 1 + extra // dec and np.abs(ys - ref[: ys.shape[0]]).max() <= 2e-6 * np.abs(ref).max() + 1e-3


@pytest.mark.parametrize("ntaps,dec,nch,ch", [
    (421, 14, 1, 0),      # configs[4]: 31 taps per phase, 3 steps of 16 columns, 42 (phase, step) pairs = 6 per wave
    (421, 14, 2, 1),      # a channel of a two-channel capture: the general staging
    (64, 16, 1, 0),       # 4 taps per phase, 2 steps
    (33, 3, 1, 0),        # 3 phases, 2 steps: one pair per wave, two waves idle
    (100, 5, 1, 0),
    (97, 4, 1, 0),
    (171, 3, 1, 0),       # 57 taps per phase: 5 steps
    (232, 8, 1, 0),       # 29 taps per phase, 8 phases
    (31, 1, 1, 0),        # no decimation: one phase
    (700, 16, 1, 0),      # 44 taps per phase, 16 phases: 64 pairs do not fit six per wave -> the vector form answers (forced or not)
])
def test_fir_matrix_core_form_shapes(ntaps, dec, nch, ch, monkeypatch):
    """k_fir_mfma (fp16 matrix cores, samples and taps split into exact fp16 pieces) forced wherever its geometry fits, against the fp64
    direct sum with the gates of the vector forms: pair counts 1..6 per wave, idle waves, both staging paths (16-byte loads with the
    next trip asked for ahead / 4-byte loads), several trips per workgroup with a ragged last one, the shortest inputs; random taps
    spanning three decades (the scaling to fp16 range and the two-piece split), full-scale samples of both signs."""
    from amaranth_twstft_amd import frontend
    monkeypatch.setenv("TWX_FIR_MFMA", "1")
    rng = np.random.default_rng(2000 * ntaps + dec)
    taps = (rng.normal(0, 1, ntaps) * np.hamming(ntaps) / np.sqrt(ntaps) * 10.0 ** rng.uniform(-3, 0, ntaps)).astype(np.float32)
    n_in = dec * 256 * 5 + 3 * ntaps + 11                                # five trips and a ragged tail
    raw = np.clip(rng.normal(0, 9000, (n_in, 2 * nch)), -32768, 32767).astype(np.int16)
    raw[7, :] = -32768; raw[8, :] = 32767; raw[9, :] = -1; raw[10, :] = 255; raw[11, :] = 256; raw[12, :] = -256      # the split's corner values
    y = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="f32")
    x = raw[:, 2 * ch].astype(np.float64) + 1j * raw[:, 2 * ch + 1]
    ref = orc.fir_decimate(x, taps.astype(np.float64), dec)
    assert y.shape == ref.shape and y.shape[0] == (n_in - ntaps) // dec + 1
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-3
    y16 = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="int16")
    assert np.abs(y16[:, 0] - np.clip(np.rint(ref.real), -32768, 32767)).max() <= 1
    assert np.abs(y16[:, 1] - np.clip(np.rint(ref.imag), -32768, 32767)).max() <= 1
    for extra in (0, dec):
        ys = frontend.fir_decimate(raw[: ntaps + extra], taps, dec, n_channels=nch, channel=ch, out="f32")
        assert ys.shape[0] == 1 + extra // dec and np.abs(ys - ref[: ys.shape[0]]).max() <= 2e-6 * np.abs(ref).max() + 1e-3
    # the two forms against each other on the same call: int16 outputs within one count, floats within the gate
    monkeypatch.setenv("TWX_FIR_MFMA", "0")
    yv = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="f32")
    yv16 = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="int16")
    assert np.abs(y - yv).max() <= 4e-6 * np.abs(ref).max() + 2e-3 and np.abs(y16.astype(np.int32) - yv16).max() <= 1


def test_wideband_chain_70msps():
    """configs[4] in miniature: chips held 28 samples at 70 Msps → FIR ↓14 → standard chain at 5 Msps;
    the lag found equals the lag the oracle finds on the same decimated int16 samples, fp32 vs fp64
    peak magnitude within 1e-6."""
    from amaranth_twstft_amd import frontend
    nchips = 10000
    chips = chips_for(14, 43, nchips)
    sps_in, dec = 28, 14
    n_in = nchips * sps_in
    taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
    p = synth.SynthParams(delay_q8=(3000 * 14 + 5) * 256, fstep=synth.fstep_for_df(500.0, 70e6), phi0=3, amp=2000,
                          noise_gain=synth.noise_gain_for_sigma(3000.0), seed=71)
    wide = synth.synth_channel(n_in + taps.size + dec, chips, sps_in, p)
    narrow = frontend.fir_decimate(wide, taps, dec, out="int16")[: 2 * nchips]
    assert narrow.shape == (2 * nchips, 2)
    with Correlator(chips, fs=FS, Nint=1) as c32, Correlator(chips, fs=FS, Nint=1, precision="f64") as c64:
        g32 = c32.process(narrow, n_channels=1, channel=0, df=500.0)[0]
        g64 = c64.process(narrow, n_channels=1, channel=0, df=500.0)[0]
    d = orc.deinterleave(narrow, 1, 0)
    d = d - d.mean()
    code = orc.make_code(chips, 2)
    o = orc.processing(d, None, None, np.arange(2 * nchips) / FS, orc.make_fcode(code), code, Nint=1, fs=FS, df=500.0)
    assert g32.indice == g64.indice == o["indice"]
    assert abs(abs(g32.xval) - abs(g64.xval)) <= MAG_TOL * abs(g64.xval)
    # y[m] is centred on input sample m*dec + (ntaps-1)/2, so the lag moves EARLIER by the half length
    expect = (3000 * 14 + 5 - (taps.size - 1) / 2) / 14.0
    assert abs(g32.indice / 3.0 - expect) < 1.0


@pytest.mark.parametrize("case", ["n2M", "n2M_loopback", "n5M_C2", "n5M_taps57_remote"])
def test_device_vs_reference_221219_processing_values(case):
    """The device path with the fine-frequency step on, against the RETURN VALUES of the reference's
    own processing() (experiments/221219_twoway/processing/godual_ranging.py:18-65) stored in
    tests/golden/ref221219_processing.json — no oracle in between."""
    g = load_golden("ref221219_processing.json")
    c = next(x for x in g["cases"] if x["name"] == case)
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    n = 2 * len(chips)
    with Correlator(chips, fs=c["fs"], Nint=c["Nint"], fine_freq=True) as cor:
        got = cor.process(raw, n_channels=1, channel=0, band=tuple(c["band_k"]) if "band_k" in c else band_numpy(c["fs"], n))[0]
    ref = c["ref"]
    assert got.indice == ref["indice"]                                   # integer lag: bit-exact
    assert abs(got.correction - ref["correction"]) <= 2e-4
    assert abs(got.df - ref["df"]) <= 1e-5                               # Hz (README parity level: 1e-3 Hz)
    delay_err = abs((got.indice + got.correction) - (ref["indice"] + ref["correction"])) / c["fs"] / 3
    assert delay_err <= 2e-11                                            # 20 ps
    for k in ("SNRr", "SNRi", "puissancecode"):
        assert abs(getattr(got, k) - ref[k]) <= 3e-4 * max(ref["SNRr"], ref["SNRi"], ref[k]) + 1e-30, k
    for k in ("puissance", "puissancenoise"):
        assert abs(getattr(got, k) - ref[k]) <= 1e-6 * ref[k], k


def test_process_file_matches_in_memory_path(tmp_path):
    """File-in / results-out (godual_ranging.m:70-103): pinned, slot-pipelined ingest == in-memory call,
    including the 2-channel layout, a byte offset skip and a short final window."""
    chips, raw = _capture(15, 3, 25000, 21, seed=91)          # 21 windows of 50000 samples, 2 channels
    n = 50000
    path = tmp_path / "1670074501.bin"
    tail = raw[:1234]
    np.concatenate([raw, tail]).tofile(path)                    # + a short final window
    band = band_numpy(FS, n)
    with Correlator(chips, fs=FS, Nint=1, max_batch=4) as cor:
        mem = cor.process(raw, n_channels=2, channel=1, band=band)
        fil = cor.process_file(str(path), n_channels=2, channel=1, band=band)
        skp = cor.process_file(str(path), n_channels=2, channel=0, df=1780.75, skip_samples=3 * n, max_windows=5)
        ref = cor.process(raw[3 * n:8 * n], n_channels=2, channel=0, df=1780.75)
    assert len(fil) == len(mem) == 21
    for a, b in zip(fil, mem):
        assert a.indice == b.indice and a.xval == b.xval and a.df == b.df and a.SNRr == b.SNRr
    assert len(skp) == 5
    for a, b in zip(skp, ref):
        assert a.indice == b.indice and a.xval == b.xval


def test_script_level_drop_in(tmp_path):
    """godual_ranging.m as a whole: directory of captures + directory of codes in → TSV rows + .mat out."""
    import io
    import scipy.io
    from amaranth_twstft_amd import godual_ranging
    chips, raw = _capture(14, 43, 10000, 3, seed=101)
    (tmp_path / "codes").mkdir()
    prn.lfsr_chips(14, 43, 10000).tofile(tmp_path / "codes" / "noiselen10000_bitlen14_taps43.bin")
    prn.lfsr_chips(14, 57, 10000).tofile(tmp_path / "codes" / "noiselen10000_bitlen14_taps57.bin")
    raw.tofile(tmp_path / "1670074501.bin")
    buf = io.StringIO()
    done = godual_ranging.run(str(tmp_path), str(tmp_path / "codes"), remote=0, OP=0, out=buf)
    assert len(done) == 1 and done[0].endswith("1670074501.mat")
    lines = buf.getvalue().split("\n")
    assert lines[1].startswith("n\tdt1\tdf1") and len([l for l in lines if l[:1].isdigit() and "\t" in l]) == 3
    m = scipy.io.loadmat(done[0])
    ref = orc.ranging(raw, chips, fs=FS, Nint=1, n_channels=2, band="godual", ddof=1)
    assert [int(v) for v in m["indice1"][0]] == [o["indice"] + 1 for o in ref[0]]       # Octave 1-based
    assert [int(v) for v in m["indice2"][0]] == [o["indice"] + 1 for o in ref[1]]
    assert np.allclose(m["SNR1r"][0], [o["SNRr"] for o in ref[0]], rtol=3e-4)
    buf2 = io.StringIO()
    assert godual_ranging.run(str(tmp_path), str(tmp_path / "codes"), out=buf2) == [] and "already done" in buf2.getvalue()


def test_edge_cases_zero_input_wraparound_and_full_scale():
    """Edge cases the reference's indexing implies: all-zero window (arg-max = first index), a peak at
    lag 0 (neighbours wrap around the circular map), full-scale int16 samples."""
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    temps = np.arange(n) / FS
    with Correlator(chips, fs=FS, Nint=1) as cor:
        # (a) zeros: every |prnmap| equal → first index; parabola 0/0 = NaN like the reference
        z = np.zeros((n, 2), dtype=np.int16)
        g = cor.process(z, n_channels=1, channel=0, df=0.0)[0]
        assert g.indice == 0 and np.isnan(g.correction) and g.puissance == 0.0
        # (b) delay 0: peak at index 0, xvalm1 is prnmap[3N-1]
        p = synth.SynthParams(delay_q8=0, fstep=0, phi0=0, amp=1000, noise_gain=synth.noise_gain_for_sigma(50.0), seed=3)
        raw = synth.synth_channel(n, chips, 2, p)
        g = cor.process(raw, n_channels=1, channel=0, df=0.0)[0]
        d = orc.deinterleave(raw, 1, 0)
        d = d - d.mean()
        o = orc.processing(d, None, None, temps, fcode, code, Nint=1, fs=FS, df=0.0)
        assert o["indice"] == 0
        _check(g, o)
        # (c) full-scale square wave following the code: |I| = 32767/32768
        fs_raw = np.empty((n, 2), dtype=np.int16)
        fs_raw[:, 0] = np.where(np.roll(code, 4242) > 0, 32767, -32768)
        fs_raw[:, 1] = np.where(np.roll(code, 4242) > 0, -32768, 32767)
        g = cor.process(fs_raw, n_channels=1, channel=0, df=0.0)[0]
        d = orc.deinterleave(fs_raw, 1, 0)
        d = d - d.mean()
        o = orc.processing(d, None, None, temps, fcode, code, Nint=1, fs=FS, df=0.0)
        _check(g, o)
        assert g.indice == 3 * 4242


# ---------------------------------------------------------------------------------------------
# acquisition stage + tracked multi-code loop (acquisition/claudio_aligned_code_ranging_separate.m)
# ---------------------------------------------------------------------------------------------

def _tracked_capture(ncodes=170, df=30.0, delay=1500, seed=3, sigma=300.0, amp=500):
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(df, FS), phi0=5, amp=amp,
                          noise_gain=synth.noise_gain_for_sigma(sigma), seed=seed)
    return chips, n, synth.synth_channel(n * ncodes, chips, 2, p)


def test_squared_spectrum_bins_and_band():
    """abs(fft(d.^2)) over a chunk longer than the code (search_df :30, carrier update :162-163)."""
    from amaranth_twstft_amd.tracked import _DevBuf
    chips, n, raw = _tracked_capture(ncodes=52)
    with Correlator(chips, fs=FS, Nint=1, convention="claudio", var_ddof=1) as cor:
        buf = _DevBuf(cor._lib, raw.nbytes)
        try:
            buf.upload(0, raw)
            d = orc.deinterleave(raw, 1, 0)
            for Ld in (50 * n, 50 * n + 1234, 51 * n + 7):          # the carry makes the chunk length arbitrary
                ref = np.fft.fft(d[:Ld] ** 2)
                bins = np.array([-3, -1, 0, 1, 2, 12, 13, Ld // 2, -(Ld // 2)])
                got = cor.sqspec_bins_dev(buf.ptr, Ld, bins)
                want = ref[bins % Ld]
                assert np.abs(got - want).max() <= 1e-9 * np.abs(ref).max()
            Lc = 50 * n
            ref = np.abs(np.fft.fft(d[:Lc] ** 2))
            k_lo, nk = -3200, 6400
            got = cor.sqspec_band_dev(buf.ptr, Lc, k_lo, nk)
            want = ref[np.arange(k_lo, k_lo + nk) % Lc]
            assert np.abs(got - want).max() <= 2e-6 * want.max()
            assert int(np.argmax(got)) == int(np.argmax(want))
        finally:
            buf.close()


def test_tracked_multicode_loop_vs_oracle():
    """search_df + per-chunk carrier + 40-ms code loop with re-alignment, against the oracle's
    restatement of claudio_aligned_code_ranging_separate.m:143-205 (unpinned: Octave only)."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    chips, n, raw = _tracked_capture()
    Lc = 50 * n
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc)
    with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc) as tr:
        got = tr.run(raw)
    assert got["kbon"] == want["kbon"] and want["kbon"] > 0
    assert got["df"] == want["df"]
    assert got["moved"] == want["moved"] == [1] and np.allclose(got["movedval"], want["movedval"])
    assert len(got["indice1"]) == len(want["indice1"]) > 100
    assert got["indice1"] == want["indice1"]                     # integer lags (and the /3 bookkeeping) bit-exact
    gx, wx = np.abs(np.array(got["xval"])), np.abs(np.array(want["xval"]))
    assert np.abs(gx - wx).max() <= MAG_TOL * wx.max()
    assert np.abs(np.array(got["correction1"]) - np.array(want["correction1"])).max() < 2e-4
    for key in ("SNR1r", "SNR1i", "puissance1"):
        a, b = np.array(got[key]), np.array(want[key])
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-12, key
    assert got["batches"] <= 2 + len(want["df"])                 # one batch per chunk once aligned


def test_tracked_loop_realigns_after_a_jump():
    """A mid-capture delay jump (sample loss) moves the window once more; codes before and after agree with the oracle."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    chips, n, a = _tracked_capture(ncodes=60, delay=1500, seed=4)
    _, _, b = _tracked_capture(ncodes=60, delay=1500 + 777, seed=5)
    raw = np.concatenate((a, b))
    Lc = 50 * n
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc)
    with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc) as tr:
        got = tr.run(raw)
    assert len(want["moved"]) >= 2
    assert got["moved"] == want["moved"] and got["indice1"] == want["indice1"]
    gx, wx = np.abs(np.array(got["xval"])), np.abs(np.array(want["xval"]))
    assert np.abs(gx - wx).max() <= MAG_TOL * wx.max()


@pytest.mark.parametrize("precision,delay", [("f32", 123456), ("f64", 123456), ("f64", 400 * 77 - 1), ("f32", 400 * 77 - 1)])
def test_tracked_loop_reference_sizes(precision, delay):
    """The script's own sizes: 100 kchip code (N = 200 000 = 500 x 400, 40 ms), 2-s chunks of 10^7 samples (:15,125-131) — in fp32
    and in fp64 (the context's precision: every kernel of the flow in complex double), the code phase also on the last element of
    a row of the two-pass layout."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    nchips, n = 100000, 200000
    chips = chips_for(17, 9, nchips)
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(12.0, FS), phi0=9, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=21)
    raw = synth.synth_channel(n * 101, chips, 2, p)
    want = orc.ranging_tracked(raw, chips, fs=FS)
    with TrackedRanging(chips, fs=FS, Nint=1, precision=precision) as tr:
        got = tr.run(raw)
    assert got["kbon"] == want["kbon"] and want["kbon"] > 0 and got["df"] == want["df"]
    assert got["moved"] == want["moved"] and got["indice1"] == want["indice1"] and len(want["indice1"]) >= 98
    gx, wx = np.abs(np.array(got["xval"])), np.abs(np.array(want["xval"]))
    assert np.abs(gx - wx).max() <= (MAG_TOL if precision == "f32" else 1e-11) * wx.max()
    assert np.abs(np.array(got["correction1"]) - np.array(want["correction1"])).max() < (2e-4 if precision == "f32" else 1e-8)


def _tracked_agrees(got, want, floor=False):
    assert got["kbon"] == want["kbon"] and got["df"] == want["df"]
    assert got["moved"] == want["moved"] and np.allclose(got["movedval"], want["movedval"])
    assert got["indice1"] == want["indice1"]                     # integer lags and the script's /3 (or floor) bookkeeping
    gx, wx = np.abs(np.array(got["xval"])), np.abs(np.array(want["xval"]))
    assert np.abs(gx - wx).max() <= MAG_TOL * wx.max()
    assert np.abs(np.array(got["correction1"]) - np.array(want["correction1"])).max() < 2e-4
    for key in ("SNR1r", "SNR1i", "puissance1"):
        a, b = np.array(got[key]), np.array(want[key])
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-12, key


@pytest.mark.parametrize("mode,OP", [("lo", 0), ("re", 0), ("re", 1)])
def test_tracked_lo_and_re_siblings_vs_oracle(mode, OP):
    """claudio_aligned_code_lo_separate.m:117-164 (carrier = full-band arg-max of every fresh chunk :126-129, floor of the
    lag :134, no search) and claudio_aligned_code_re_separate.m (search_df in the remote band :137-141) through
    twx_tracked_host, against the oracle's restatement of those scripts (unpinned: Octave only)."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    m = orc.tracked_mode(mode, OP)
    car = 1234.5 if mode == "lo" else (m["band"][0] + m["band"][1]) / 4
    chips, n, a = _tracked_capture(ncodes=60, df=car, delay=1500, seed=14)
    _, _, b = _tracked_capture(ncodes=60, df=car, delay=1500 + 333, seed=15)
    raw = np.concatenate((a, b))
    Lc = 50 * n
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, band=m["band"], carrier=m["carrier"], indice_floor=m["indice_floor"])
    with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc, mode=mode, OP=OP) as tr:
        got = tr.run(raw)
    assert len(want["indice1"]) >= 100 and len(want["moved"]) >= 2 and (want["kbon"] > 0) == (mode != "lo")
    _tracked_agrees(got, want)
    if mode == "lo":
        assert all(float(v).is_integer() for v in got["indice1"])


def test_tracked_file_entry_skip_and_search_df(tmp_path):
    """twx_tracked_file: the capture FILE in, records out (what the MEX gateway calls) — the 30-s skip of :128 only moves the
    chunk search_df sees, the file is then re-read from its start (:156-159); twx_tracked_search_df alone; an unreadable
    path and a capture shorter than one chunk."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    from amaranth_twstft_amd import _lib as L
    chips, n, raw = _tracked_capture(ncodes=130, seed=33)
    Lc = 50 * n
    path = tmp_path / "cap_2.bin"
    raw.tofile(path)
    with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc) as tr:
        for skip in (0.0, 30 * n / FS):
            want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, skip_samples=int(skip * FS))
            got = tr.run_file(str(path), skip_seconds=skip)
            assert want["kbon"] > 0 and len(want["indice1"]) >= 95
            _tracked_agrees(got, want)
        assert tr.search_df(raw[:2 * Lc]) == want["kbon"]
        assert tr.run(raw.reshape(-1)[:Lc])["indice1"] == []       # half a chunk: no codes, no error
        with pytest.raises(L.TwxError) as e:
            tr.run_file(str(tmp_path / "missing.bin"))
        assert "cannot open" in str(e.value)
        assert tr.default_skip_samples == 30 * 5_000_000           # fseek(f,30*fs*2*2)


def test_tracked_mex_gateway_runs_the_whole_flow(tmp_path):
    """mexFunction() of mex/twstft_tracked_mex.cpp executed on the GPU box (functional fake mex.h): capture file + code bytes
    + mode in, the script's workspace variables out (1-based kbon), equal to the ctypes path record for record."""
    import subprocess
    from tests.test_abi_and_host import build_mex_harness
    from tests.test_gpu_configs import _read_mex_outputs
    from amaranth_twstft_amd.tracked import TrackedRanging
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = build_mex_harness(root, tmp_path, "mex_tracked_harness", "twstft_tracked_mex")
    nchips, n = 100000, 200000                                     # the script's own sizes: 2-s chunks of 10^7 samples
    chips = chips_for(17, 9, nchips)
    p = synth.SynthParams(delay_q8=4321 * 256, fstep=synth.fstep_for_df(-17.0, FS), phi0=9, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(500.0), seed=23)
    raw = synth.synth_channel(n * 101, chips, 2, p)
    raw.tofile(tmp_path / "cap_2.bin")
    chips.tofile(tmp_path / "n0.bin")
    for mode in ("ranging", "lo"):
        r = subprocess.run([str(exe), str(tmp_path / "cap_2.bin"), str(tmp_path / "n0.bin"), str(tmp_path / "out.bin"), mode, "0", "5e6", "1", "0"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        o = _read_mex_outputs(tmp_path / "out.bin")
        with TrackedRanging(chips, fs=FS, Nint=1, mode=mode) as tr:
            ref = tr.run(raw, skip_samples=0)
        nc = len(ref["indice1"])
        assert nc >= 98 and all(x.shape == (1, nc) for x in o[:6]) and o[6].shape == (1, len(ref["df"]))
        assert list(o[0][0]) == ref["xval"] and list(o[1][0]) == ref["indice1"] and list(o[2][0]) == ref["correction1"]
        assert list(o[3][0]) == ref["SNR1r"] and list(o[5][0]) == ref["puissance1"] and list(o[6][0]) == ref["df"]
        assert list(o[7][0]) == ref["moved"] and list(o[8][0]) == ref["movedval"] and o[9][0, 0] == ref["kbon"] + 1
        assert o[10][0, 0] == ref["puissancecode"] and o[11][0, 0] == ref["puissancenoise"]
    want = orc.ranging_tracked(raw, chips, fs=FS, band=(-20000.0, 20000.0), carrier="chunk_band", indice_floor=True)
    assert ref["indice1"] == want["indice1"] and ref["df"] == want["df"]      # the `lo` flow at the reference's sizes


def test_plain_c_client_of_the_abi(tmp_path):
    """A C99 program (no Python, no torch) drives the library end to end: tests/cpu/abi_smoke.c."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "amaranth_twstft_amd")
    exe = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpu", "abi_smoke.c"),
                    "-L" + libdir, "-ltwstft_hip", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), "10000", "14", "43", "4321"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "indice0=12963" in out.stdout
    assert "tracked(lo): codes=" in out.stdout and "moved=1 first_moved_p=1" in out.stdout      # twx_tracked_* from plain C


def test_stockham_row_pass_fallback(monkeypatch):
    """TWX_ROWD=0 selects the Stockham row kernels (the form used by plans without equal inner radices)."""
    chips, raw = _capture(17, 15, 100000, 2, seed=77)
    n = 200000
    band = band_numpy(FS, n)
    with Correlator(chips, fs=FS, Nint=0) as cor:                 # default: the DIF/DIT CAF row pass (k_rowd_caf, N2 = 4000)
        pk_d, lag_d = cor.caf_bins(raw[:n], -70, 70, n_channels=2, channel=0)
    monkeypatch.setenv("TWX_ROWD", "0")
    with Correlator(chips, fs=FS, Nint=1) as cor:
        got = cor.ranging(raw, n_channels=2, channels=(0,), band=band)
    with Correlator(chips, fs=FS, Nint=0) as cor:                 # Stockham CAF row pass (k_row_caf) on the same window
        pk_s, lag_s = cor.caf_bins(raw[:n], -70, 70, n_channels=2, channel=0)
    ref = orc.ranging(raw, chips, fs=FS, Nint=1, n_channels=2, channels=(0,), band="numpy")
    for g, o in zip(got[0], ref[0]):
        _check(g, o)
    assert np.array_equal(lag_d, lag_s) and np.abs(pk_d - pk_s).max() <= MAG_TOL * pk_s.max()
    d = orc.deinterleave(raw[:n], 2, 0)
    d = d - d.mean()
    ks, pk_o, lag_o = orc.caf_bins_shift(d, orc.make_fcode(orc.make_code(chips, 2)), -70, 70)
    assert np.array_equal(lag_d, lag_o) and np.abs(pk_d - pk_o).max() <= MAG_TOL * pk_o.max()


@pytest.mark.parametrize("bitlen,taps,nchips,remote", [(19, 39, 524288, 0), (22, 3, 2_500_000, 0), (22, 3, 2_500_000, 1), (22, 3, 1_250_000, 0)])
def test_band_row_pass_two_forms_agree(monkeypatch, bitlen, taps, nchips, remote):
    """The carrier search's row pass in fp32 with a narrow band runs k_rowd_bandsum (sums over the workgroup's threads, no row in LDS;
    rows of 4096 (a plug-in plan, R0 = 16), 8000 and 4000 points); TWX_BANDSUM=0 selects k_rowd<BAND> with its pruned last stage.  Same peak bin for every window — a tone
    inside the near band (godual_ranging.m:83-84) and inside the remote band (:86-89, other digit pairs) — hence byte-identical records,
    and the bin is the oracle's."""
    import torch
    lib = L.load()
    dev = torch.device("cuda", 0)
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    nwin = 6
    iq = torch.empty((nwin, n, 2), dtype=torch.int16, device=dev)
    cd = torch.from_numpy(chips).to(dev)
    dfs = [(50_000.0 - 1234.25 * w) if remote else (9000.5 - 3100.0 * w) for w in range(nwin)]      # 2 df inside the band in both cases
    for w in range(nwin):
        p = synth.SynthParams(delay_q8=(4321 + w) * 256, fstep=synth.fstep_for_df(dfs[w], FS), phi0=w, amp=300,
                              noise_gain=synth.noise_gain_for_sigma(300.0), seed=50 + w)
        params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(iq[w].data_ptr(), n, 0, cd.data_ptr(), nchips, 2, 1, params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    band = L.twx_band(*band_godual(FS, n, remote=remote))
    RB = C.sizeof(L.twx_result)
    out = {}
    with Correlator(chips, fs=FS, Nint=1) as cor:
        for form in ("1", "0"):
            monkeypatch.setenv("TWX_BANDSUM", form)
            res = torch.zeros((nwin, RB), dtype=torch.uint8, device=dev)
            L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
            cor.synchronize()
            out[form] = res.cpu().numpy()
    assert out["1"].tobytes() == out["0"].tobytes()
    recs = (L.twx_result * nwin).from_buffer_copy(out["1"].tobytes())
    for w in range(nwin):
        assert abs(recs[w].df - dfs[w]) <= 0.51 * FS / n, (w, recs[w].df, dfs[w])      # df = half the d^2 peak's frequency: 1-bin grid of fs/n
        assert int(recs[w].indice0) == 3 * (4321 + w)
    if nchips <= 524288:                                             # the oracle's own search (fft(d.^2) over the band) on every window
        freq = orc.freq_axis(FS, n)
        k = orc.band_godual(freq, remote=remote)
        for w in range(nwin):
            d = orc.deinterleave(iq[w].cpu().numpy().reshape(-1), 1, 0)
            assert abs(recs[w].df - orc.coarse_df(d - d.mean(), k, freq)[1]) < 1e-9


@pytest.mark.parametrize("bitlen,taps,nchips", [(22, 3, 2_500_000), (22, 3, 1_250_000)])
def test_band_row_pass_two_forms_agree_on_random_bands(monkeypatch, bitlen, taps, nchips):
    """Random search bands — one bin wide to 200 kHz wide, anywhere on the axis, around a tone or over noise only — through both forms of the
    row pass: up to 8 digit pairs take k_rowd_bandsum, wider bands fall back to k_rowd<BAND> inside the library, and the records agree byte
    for byte either way (noise-only windows included: the arg-max of a few hundred thousand noise bins)."""
    import torch
    lib = L.load()
    dev = torch.device("cuda", 0)
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    nwin = 4
    rng = np.random.default_rng(int(os.environ.get("TWX_SWEEP_SEED", "0")) + nchips)
    iq = torch.empty((nwin, n, 2), dtype=torch.int16, device=dev)
    cd = torch.from_numpy(chips).to(dev)
    tones = [float(rng.uniform(-9e4, 9e4)) for _ in range(nwin)]
    for w in range(nwin):
        p = synth.SynthParams(delay_q8=(777 + w) * 256, fstep=synth.fstep_for_df(tones[w], FS), phi0=w, amp=300 if w != 2 else 0,
                              noise_gain=synth.noise_gain_for_sigma(300.0), seed=90 + w)
        params = np.array([p.delay_q8, p.fstep, p.phi0, p.amp, p.noise_gain, p.seed, p.stream, 0], dtype=np.int64)
        L.check(lib.twx_synth_capture_dev(iq[w].data_ptr(), n, 0, cd.data_ptr(), nchips, 2, 1, params.ctypes.data_as(C.c_void_p), None))
    torch.cuda.synchronize()
    RB = C.sizeof(L.twx_result)
    ntrial = 24
    with Correlator(chips, fs=FS, Nint=1) as cor:
        for t in range(ntrial):
            width = int(10 ** rng.uniform(0, np.log10(0.04 * n)))                      # bins of the fftshifted axis
            centre = n // 2 + int(round(2 * tones[t % nwin] * n / FS)) if t % 3 else int(rng.integers(width, n - width))
            lo = int(np.clip(centre - width // 2, 0, n - 1)); hi = int(np.clip(lo + width, lo, n - 1))
            band = L.twx_band(lo, hi)
            out = {}
            for form in ("1", "0"):
                monkeypatch.setenv("TWX_BANDSUM", form)
                res = torch.zeros((nwin, RB), dtype=torch.uint8, device=dev)
                L.check(lib.twx_process_windows_dev(cor._h, iq.data_ptr(), nwin, 1, 0, C.byref(band), None, res.data_ptr()), cor._h)
                cor.synchronize()
                out[form] = res.cpu().numpy()
            assert out["1"].tobytes() == out["0"].tobytes(), (t, lo, hi)


def test_host_pipeline_many_chunks_per_window_df():
    """twx_process_windows through the pinned pipeline: more chunks than slots, a ragged tail, per-window df."""
    nchips, n, nwin = 10000, 20000, 64 * 3 + 64 + 5              # batch 64: 4 full chunks + a 5-window tail
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=777 * 256, fstep=synth.fstep_for_df(1500.0, FS), phi0=3, amp=400,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=11)
    raw = synth.synth_channel(n * nwin, chips, 2, p)
    dfs = np.where(np.arange(nwin) % 3 == 0, 1500.0, 1500.0 + 40.0 * (np.arange(nwin) % 5))
    with Correlator(chips, fs=FS, Nint=1, max_batch=64) as cor:
        assert cor.info.batch == 64
        got = cor.process(raw, 1, 0, df=dfs)
        assert len(got) == nwin
        for w in (0, 1, 63, 64, 65, 191, 192, 255, 256, nwin - 1):
            one = cor.process(raw[w * n:(w + 1) * n], 1, 0, df=float(dfs[w]))[0]
            g = got[w]
            assert g.indice == one.indice and g.xval == one.xval and g.correction == one.correction and g.df == dfs[w]
        # windows mixed with the true offset see the full peak, the others a weaker one at the same lag
        strong = [abs(got[w].xval) for w in range(nwin) if dfs[w] == 1500.0]
        assert min(strong) > 0 and all(got[w].indice == 3 * 777 for w in range(nwin) if dfs[w] == 1500.0)


@pytest.mark.parametrize("snr_db", [-25.0, -20.0, -10.0])
def test_wipeoff_snr_tracks_the_known_snr(snr_db):
    """The property experiments/221127_SNR/interpolation_effect.m demonstrates (README table :43-51): the code
    wipe-off estimate mean(y.*code)^2/var(y.*code) follows the known signal-to-noise ratio of a synthetic signal.
    With the reference's rotate offset of -1 grid sample (godual_ranging.m:43) the estimate sits 0.4-0.7 dB low for
    Nint = 1, 2 (the band-limited interpolation of a 2-samples-per-chip code) and 6 dB low for Nint = 0, where the
    offset is a whole sample = half a chip; the reference only ever runs this formula with Nint = 1.  Above ~0 dB the
    misaligned product's own variance dominates and the estimate saturates (+6.8 dB at a true +10 dB), so the check
    covers the operating range of the link (README.md(221219):47: -9 dB)."""
    nchips, n = 100000, 200000
    chips = chips_for(17, 9, nchips)
    amp = 300.0
    sigma = amp / np.sqrt(2.0 * 10 ** (snr_db / 10))              # SNR = A^2 / (2 sigma^2): complex noise variance 2 sigma^2
    p = synth.SynthParams(delay_q8=4321 * 256, fstep=0, phi0=123456789, amp=int(amp),
                          noise_gain=synth.noise_gain_for_sigma(float(sigma)), seed=31)
    raw = synth.synth_channel(n, chips, 2, p)
    est = {}
    for nint in (0, 1, 2):
        with Correlator(chips, fs=FS, Nint=nint) as cor:
            r = cor.process(raw, 1, 0, df=0.0)[0]
        assert r.indice == (2 * nint + 1) * 4321
        est[nint] = 10 * np.log10(r.SNRr + r.SNRi)
    assert -0.9 < est[1] - snr_db < 0.1 and -1.1 < est[2] - snr_db < 0.1, est
    assert abs(est[0] - (snr_db - 6.0)) < 0.6, est


@pytest.mark.parametrize("variant", ["besancon", "qpsk"])
def test_replica_variants_of_the_experiment_scripts(variant):
    """Zero-mean 0/1 code (experiments/220616_Besancon/godual.m:5-8) and complex QPSK code
    (experiments/220822_qpsk_vs_bpsk/goqpsk.m:10-16), both with ``prnmap=ifft(fft(y).*fcode)`` (Nint = 0)."""
    nchips, n = 10000, 20000
    ci = chips_for(14, 43, nchips)
    cq = chips_for(14, 57, nchips) if variant == "qpsk" else None
    p = synth.SynthParams(delay_q8=2345 * 256, fstep=synth.fstep_for_df(900.0, FS), phi0=17, amp=400,
                          noise_gain=synth.noise_gain_for_sigma(250.0), seed=13)
    raw = synth.synth_channel(n, ci, 2, p)
    code = orc.make_code_variant(ci, cq, 2, unipolar=True, zero_mean=True)
    fcode = np.conj(np.fft.fft(code))
    with Correlator(ci, fs=FS, Nint=0, chips_q=cq, code_levels="unipolar", code_zero_mean=True) as cor:
        cs = cor.code_spectrum()
        assert np.abs(cs - fcode).max() <= 2e-7 * np.abs(fcode).max() and abs(cs[0]) <= 1e-6 * np.abs(fcode).max()
        g = cor.process(raw, 1, 0, df=900.0)[0]
        zmap = cor.xcorr_map(raw, 900.0)
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    y = d * np.exp(-2j * np.pi * 900.0 * np.arange(n) / FS)
    ref = np.fft.ifft(np.fft.fft(y) * fcode)
    assert np.abs(zmap - ref).max() <= 2e-6 * np.abs(ref).max()
    ind = int(np.abs(ref).argmax())
    assert g.indice == ind == 2345
    assert abs(abs(g.xval) - abs(ref[ind])) <= MAG_TOL * abs(ref[ind])
    assert np.isnan(g.SNRr) and np.isnan(g.puissancenoise) and g.puissance > 0          # wipe-off statistics undefined here


def test_randomised_parity_sweep():
    """Seeded sweep over code lengths, delays, carrier offsets and SNRs down to noise-dominated maps (where the
    arg-max is decided among noise peaks): integer lag and carrier bin identical to the oracle in every case."""
    rng = np.random.default_rng(20260101)
    cases = [(13, 27, 5000), (14, 57, 10000), (15, 17, 25000)]
    mism = []
    for bitlen, taps, nchips in cases:
        chips = chips_for(bitlen, taps, nchips)
        n = 2 * nchips
        code = orc.make_code(chips, 2)
        fcode = orc.make_fcode(code)
        freq = orc.freq_axis(FS, n)
        k = orc.band_numpy(freq)
        temps = np.arange(n) / FS
        band = band_numpy(FS, n)
        nwin = int(os.environ.get("TWX_SWEEP_WINDOWS", "24"))       # raise for a longer hunt
        params = []
        raws = []
        for w in range(nwin):
            amp = int(rng.choice([0, 20, 60, 200, 1000]))
            sigma = float(rng.choice([50.0, 300.0, 2000.0]))
            df = float(rng.uniform(-7000, 7000))
            p = synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256 + int(rng.integers(0, 256)),
                                  fstep=synth.fstep_for_df(df, FS), phi0=int(rng.integers(0, 2 ** 32)), amp=amp,
                                  noise_gain=synth.noise_gain_for_sigma(sigma), seed=int(rng.integers(1, 10 ** 6)))
            raws.append(synth.synth_channel(n, chips, 2, p))
            params.append((amp, sigma, df))
        raw = np.concatenate(raws)
        with Correlator(chips, fs=FS, Nint=1) as cor:
            got = cor.process(raw, 1, 0, band=band)
        for w, g in enumerate(got):
            d = orc.deinterleave(raws[w], 1, 0)
            d = d - d.mean()
            o = orc.processing(d, k, freq, temps, fcode, code, Nint=1, fs=FS)
            if g.indice != o["indice"] or abs(g.df - o["df"]) > 1e-9:
                mism.append((nchips, w, params[w], g.indice, o["indice"], g.df, o["df"]))
            else:
                assert abs(abs(g.xval) - abs(o["xval"])) <= MAG_TOL * abs(o["xval"])
    assert not mism, mism


def test_randomised_option_sweep():
    """Seeded sweep over the OPTIONS of processing(): interpolation factor, variance convention, the wipe-off rotation, the
    Hamming-windowed replica, fp32 / fp64, carrier searched over one of the three bands or supplied, one- and two-channel frames,
    strong to noise-dominated windows — the combinations the single-option tests above do not meet — against the oracle with the
    same settings: integer lag and carrier exact, the rest within the tolerances of _check.  TWX_SWEEP_OPTIONS raises the count."""
    rng = np.random.default_rng(424242 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "16"))
    codes = [(13, 27, 5000, 2), (14, 57, 10000, 2), (13, 27, 5000, 1), (13, 27, 5000, 4), (13, 27, 2500, 4), (14, 57, 10000, 1)]     # chips x samples per chip
    seen = set()
    for it in range(ncomb):
        bitlen, taps, nchips, sps = codes[int(rng.integers(0, len(codes)))]
        chips = chips_for(bitlen, taps, nchips)
        n = sps * nchips
        Nint = int(rng.choice([0, 1, 1, 2]))
        ddof = int(rng.integers(0, 2))
        snr_rot = int(rng.choice([-1, -1, -2, 0]))
        window = str(rng.choice(["none", "none", "hamming"]))
        precision = str(rng.choice(["f32", "f32", "f64"]))
        nch = int(rng.integers(1, 3))
        ch = int(rng.integers(0, nch))
        mode = str(rng.choice(["band_numpy", "band_godual", "band_remote", "df"]))
        seen.add((Nint, ddof, snr_rot, window, precision, nch, mode, sps))
        df_true = float(rng.uniform(-6000, 6000)) if mode != "band_remote" else float(rng.uniform(41000, 59000))     # the bands search 2*df
        nwin = 3
        chans = [synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256 + int(rng.integers(0, 256)), fstep=synth.fstep_for_df(df_true + 3.0 * c, FS),
                                   phi0=int(rng.integers(0, 2 ** 32)), amp=int(rng.choice([0, 40, 300, 2000])),
                                   noise_gain=synth.noise_gain_for_sigma(float(rng.choice([60.0, 500.0, 2500.0]))), seed=int(rng.integers(1, 10 ** 6)), stream=c)
                 for c in range(nch)]
        raw = synth.synth_capture(n * nwin, chips, sps, chans)
        code = orc.make_code(chips, sps)
        fcode = orc.make_fcode(code, "hamming" if window == "hamming" else "godual")
        freq = orc.freq_axis(FS, n)
        temps = np.arange(n) / FS
        if mode == "band_numpy":
            k, band = orc.band_numpy(freq), band_numpy(FS, n)
        elif mode == "band_godual":
            k, band = orc.band_godual(freq), band_godual(FS, n)
        elif mode == "band_remote":
            k, band = orc.band_godual(freq, 1, 0), band_godual(FS, n, remote=1, OP=0)
        else:
            k = band = None
        dfs = [df_true + 0.37 * w for w in range(nwin)]
        # a third of the combinations hand over what the reference's own call gets: the complex column d, mean already removed by the
        # caller (godual_ranging.m:80), here with a gain and a rotation so that it is no longer integer-valued (twx_process_complex)
        cplx = bool(rng.integers(0, 3) == 0)
        gain = (0.37 + 0.11j) if cplx else 1.0
        wins = []
        for w in range(nwin):
            d = orc.deinterleave(raw[w * n:(w + 1) * n], nch, ch)
            wins.append((d - d.mean()) * gain)
        with Correlator(chips, fs=FS, sps=sps, Nint=Nint, var_ddof=ddof, snr_rot=snr_rot, window=window, precision=precision, max_batch=int(rng.integers(1, 4))) as cor:
            if cplx:
                got = cor.processing_complex(np.concatenate(wins), k=band) if band is not None else cor.processing_complex(np.concatenate(wins), df=dfs)
            else:
                got = cor.process(raw, nch, ch, band=band) if band is not None else cor.process(raw, nch, ch, df=dfs)
        assert len(got) == nwin
        for w, g in enumerate(got):
            d = wins[w]
            o = orc.processing(d, k, freq, temps, fcode, code, Nint=Nint, fs=FS, snr_rot=snr_rot, ddof=ddof, df=None if band is not None else dfs[w])
            try:
                _check(g, o)
            except AssertionError as e:
                raise AssertionError(f"combination {it}: {nchips} chips x {sps} Nint={Nint} ddof={ddof} snr_rot={snr_rot} window={window} {precision} nch={nch} ch={ch} {mode} complex_input={cplx} window {w}: {e}") from e
    assert len(seen) >= min(ncomb, 12)


def test_randomised_option_sweep_conventions_and_replicas():
    """The second option sweep: the claudio convention (fcode.*conj(ffty), Octave variances) against processing_claudio, the replica
    variants of the experiment scripts (0/1 levels, zero mean, complex QPSK replica) against a direct ifft(fft(y).*fcode), and the
    all-channel call (both channels of a two-channel capture from one copy) — each with random interpolation factor, precision, batch
    size and signal level.  TWX_SWEEP_OPTIONS raises the count."""
    rng = np.random.default_rng(777 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "16"))
    codes = [(13, 27, 5000), (14, 57, 10000)]
    for it in range(ncomb):
        bitlen, taps, nchips = codes[int(rng.integers(0, len(codes)))]
        chips = chips_for(bitlen, taps, nchips)
        n = 2 * nchips
        Nint = int(rng.choice([0, 1, 1, 2]))
        R = 2 * Nint + 1
        precision = str(rng.choice(["f32", "f32", "f64"]))
        kind = str(rng.choice(["claudio", "claudio", "unipolar", "zero_mean", "qpsk", "all_channels"]))
        nch = 2 if kind == "all_channels" else int(rng.integers(1, 3))
        ch = int(rng.integers(0, nch))
        nwin = 3
        df0 = float(rng.uniform(-5000, 5000))
        dfs = [df0 + 0.61 * w for w in range(nwin)]
        chans = [synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256 + int(rng.integers(0, 256)), fstep=synth.fstep_for_df(df0, FS),
                                   phi0=int(rng.integers(0, 2 ** 32)), amp=int(rng.choice([0, 60, 400, 2500])),
                                   noise_gain=synth.noise_gain_for_sigma(float(rng.choice([60.0, 500.0, 2500.0]))), seed=int(rng.integers(1, 10 ** 6)), stream=c)
                 for c in range(nch)]
        raw = synth.synth_capture(n * nwin, chips, 2, chans)
        temps = np.arange(n) / FS
        mb = int(rng.integers(1, 4))
        tag = f"combination {it}: {kind} Nint={Nint} {precision} nch={nch} ch={ch} max_batch={mb}"
        win = lambda w, c: (lambda d: d - d.mean())(orc.deinterleave(raw[w * n:(w + 1) * n], nch, c))
        try:
            if kind == "claudio":
                ddof = int(rng.integers(0, 2))
                code = orc.make_code(chips, 2)
                fc = orc.make_fcode(code, "claudio")
                with Correlator(chips, fs=FS, Nint=Nint, convention="claudio", var_ddof=ddof, precision=precision, max_batch=mb) as cor:
                    got = cor.process(raw, nch, ch, df=dfs)
                for w, g in enumerate(got):
                    _check(g, orc.processing_claudio(win(w, ch), dfs[w], temps, fc, code, Nint=Nint, ddof=ddof))
            elif kind == "all_channels":
                code = orc.make_code(chips, 2)
                fcode = orc.make_fcode(code)
                df2 = np.array([[d, d - 2.5] for d in dfs])
                with Correlator(chips, fs=FS, Nint=Nint, precision=precision, max_batch=mb) as cor:
                    got = cor.process(raw, 2, -1, df=df2)
                for c in (0, 1):
                    assert len(got[c]) == nwin
                    for w, g in enumerate(got[c]):
                        _check(g, orc.processing(win(w, c), None, None, temps, fcode, code, Nint=Nint, fs=FS, df=float(df2[w, c])))
            else:
                cq = chips_for(14, 57, nchips) if kind == "qpsk" else None
                unipolar = kind in ("unipolar", "qpsk") or bool(rng.integers(0, 2))
                zero_mean = kind in ("zero_mean", "qpsk") or bool(rng.integers(0, 2))
                code = orc.make_code_variant(chips, cq, 2, unipolar=unipolar, zero_mean=zero_mean)
                fcode = np.conj(np.fft.fft(code))
                with Correlator(chips, fs=FS, Nint=Nint, chips_q=cq, code_levels="unipolar" if unipolar else "bipolar", code_zero_mean=zero_mean,
                                precision=precision, max_batch=mb) as cor:
                    got = cor.process(raw, nch, ch, df=dfs)
                for w, g in enumerate(got):
                    y = win(w, ch) * np.exp(-2j * np.pi * dfs[w] * temps)
                    z = orc.xcorr_interp(np.fft.fft(y), fcode, Nint)
                    ind, corr, xval, xm1, xp1 = orc.peak_refine(z)
                    assert g.indice == ind
                    assert abs(g.xval - xval) <= 2 * MAG_TOL * abs(xval) and abs(g.xvalm1 - xm1) <= 2 * MAG_TOL * abs(xval) and abs(g.xvalp1 - xp1) <= 2 * MAG_TOL * abs(xval)
                    assert abs(g.correction - corr) <= 2e-4 and abs(g.puissance - np.var(y)) <= 1e-6 * np.var(y)
                    if unipolar or zero_mean or cq is not None:
                        assert np.isnan(g.SNRr) and np.isnan(g.puissancecode)
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from e


@pytest.mark.parametrize("bitlen,taps,nchips", [(13, 27, 5000), (14, 57, 10000), (15, 17, 25000), (16, 45, 32768), (17, 9, 100000), (18, 39, 262144),
                                                # the plug-in lengths of __graft_entry__.PLUGIN_LENGTHS: N = 5000, 25000, 4000, 80000, 81000, 14000, 6000, 12000, 18000
                                                (13, 27, 2500), (15, 3, 12500), (12, 83, 2000), (16, 45, 40000), (16, 45, 40500), (13, 27, 7000), (12, 83, 3000),
                                                (13, 27, 6000), (14, 43, 9000)])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_whole_correlation_map_every_row_form(bitlen, taps, nchips, precision):
    """EVERY lag of the interpolated correlation map (twx_xcorr_map), not only the peak and its neighbours, against the oracle's
    ifft — over the row-pass forms the window lengths select: N2 = 400 = 20*20 (R0 = 1: k_rowd_small in fp32, the unfolded
    k_rowd in fp64), 256 = 16*16 and 4096 = 16^3 (power-of-two plug-ins), 8000 = 20^3 (the folded, row-walking form) — in both
    precisions.  (The unfolded fp64 form once wrote element N2-1 of every row wrong: idle lanes of the last wave raced the store of
    lane M-1 with another twiddle; the peak-only tests met that column with probability 1/400.)"""
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    p = synth.SynthParams(delay_q8=(n // 3 + 11) * 256 + 77, fstep=synth.fstep_for_df(0.0, FS), phi0=77, amp=900,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=nchips)
    raw = synth.synth_channel(n, chips, 2, p)
    x = orc.deinterleave(raw, 1, 0)
    x = x - x.mean()
    fx = np.fft.fft(x)
    tol = 2e-6 if precision == "f32" else 1e-12
    for Nint in (0, 1):
        with Correlator(chips, fs=FS, Nint=Nint, precision=precision) as cor:
            z = cor.xcorr_map(raw, 0.0, n_channels=1, channel=0)
            n2 = int(cor.info.n2)
        zr = orc.xcorr_interp(fx, fcode, Nint)
        err = np.abs(z - zr)
        bad = np.nonzero(err > tol * np.abs(zr).max())[0]
        assert bad.size == 0, (n, n2, precision, Nint, bad.size, sorted(set(((bad // (2 * Nint + 1)) % n2).tolist()))[:8], float(err.max() / np.abs(zr).max()))


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("bitlen,taps,nchips", [(13, 27, 5000), (14, 57, 10000), (16, 45, 32768), (17, 9, 100000)])
def test_carrier_search_finds_a_tone_on_every_bin_of_the_band(bitlen, taps, nchips, precision):
    """The coarse carrier estimate (arg-max of fftshift(abs(fft(d.^2))) over the band k, godual_ranging.m:14-15) bin by bin: one
    window per bin of the search band, each a tone whose square lands exactly on that bin — every bin must be found, i.e. every
    element of the band the pruned row pass (k_rowd<BAND> / k_row<BAND>) evaluates is right, in both precisions and over the row
    forms (N2 = 400, 256, 4000).  The peak-only tests meet a given bin with the probability of the carrier they draw."""
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    freq = orc.freq_axis(FS, n)
    k = orc.band_godual(freq)
    band = band_godual(FS, n)
    bins = k if k.size <= 600 else k[:: k.size // 600 + 1]                       # 600 windows at most: every bin for the short codes, a comb over the long
    if k.size > 600:
        bins = np.unique(np.concatenate((bins, k[:40], k[-40:], k[k.size // 2 - 20:k.size // 2 + 20])))
    bins = bins[bins != n // 2]                                                 # the tone on bin 0 is a constant: the mean removal takes it away
    t = np.arange(n)
    raw = np.empty((bins.size * n, 2), dtype=np.int16)
    for i, kb in enumerate(bins):
        kappa = int(kb) - n // 2                                                 # signed bin of the fft (fftshift: index n/2 is bin 0)
        ph = (kappa * t % (2 * n)).astype(np.float64) * (np.pi / n)               # the tone at kappa/2 bins: its square sits on bin kappa
        raw[i * n:(i + 1) * n, 0] = np.rint(8000 * np.cos(ph))
        raw[i * n:(i + 1) * n, 1] = np.rint(8000 * np.sin(ph))
    with Correlator(chips, fs=FS, Nint=0, precision=precision) as cor:
        got = cor.process(raw, 1, 0, band=band)
    assert len(got) == bins.size
    wrong = []
    for i, (g, kb) in enumerate(zip(got, bins)):
        d = orc.deinterleave(raw[i * n:(i + 1) * n], 1, 0)
        d = d - d.mean()
        idx, df = orc.coarse_df(d, k, freq)
        if abs(g.df - df) > 1e-9 or g.df_index != idx:
            wrong.append((int(kb), g.df_index, idx, g.df, df))
    assert not wrong, (len(wrong), wrong[:10])


def test_randomised_fir_and_sliding_shapes():
    """Random shapes of the two kernels beside the FFT chain against their fp64 definitions: the FIR decimator over tap count 1..1024,
    decimation 1..16, one / two channels, lengths from one output up to several workgroups with ragged tails (both kernel forms, every
    step-group count, the generic loop); the sliding dot product over period, code count, lag window 0..31, channel layout, start
    offset, carrier and phase (the three tiles and their fall-backs).  TWX_SWEEP_OPTIONS raises the count."""
    from amaranth_twstft_amd import frontend, tracking
    rng = np.random.default_rng(31337 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "16"))
    for it in range(ncomb):
        # ---- FIR
        dec = int(rng.integers(1, 17))
        ntaps = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), rng.integers(400, 1025)]))
        nch = int(rng.integers(1, 3)); ch = int(rng.integers(0, nch))
        nout = int(rng.choice([1, 2, rng.integers(3, 600), rng.integers(600, 5000)]))
        n_in = (nout - 1) * dec + ntaps + int(rng.integers(0, dec))
        taps = (rng.normal(0, 1, ntaps) / np.sqrt(ntaps)).astype(np.float32)
        raw = np.clip(rng.normal(0, 5000, (n_in, 2 * nch)), -32768, 32767).astype(np.int16)
        x = raw[:, 2 * ch].astype(np.float64) + 1j * raw[:, 2 * ch + 1]
        ref = orc.fir_decimate(x, taps.astype(np.float64), dec)
        tag = f"combination {it}: FIR ntaps={ntaps} dec={dec} nch={nch} ch={ch} nout={nout}"
        y = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="f32")
        assert y.shape == ref.shape == (nout,), tag
        # fp32 accumulation over ntaps terms of random sign: the rounding grows with sqrt(ntaps) (1011 taps: 5e-6 of the output seen)
        assert np.abs(y - ref).max() <= max(3e-6, 4e-7 * np.sqrt(ntaps)) * np.abs(ref).max() + 2e-3, (tag, float(np.abs(y - ref).max()), float(np.abs(ref).max()))
        y16 = frontend.fir_decimate(raw, taps, dec, n_channels=nch, channel=ch, out="int16")
        assert np.abs(y16[:, 0] - np.clip(np.rint(ref.real), -32768, 32767)).max() <= 1 and np.abs(y16[:, 1] - np.clip(np.rint(ref.imag), -32768, 32767)).max() <= 1, tag
        # ---- sliding dot product
        nlag = int(rng.choice([0, 1, 4, 8, 9, 16, 28, 31, rng.integers(0, 32)]))
        nobs = int(rng.choice([rng.integers(1, 64), 8 * rng.integers(1, 400), rng.integers(64, 3000), 8 * rng.integers(2000, 3000), 16384 + 8 * rng.integers(0, 4)]))
        ncodes = int(rng.integers(1, 6))
        nch = int(rng.integers(1, 3)); ch = int(rng.integers(0, nch)); pt = int(rng.integers(0, 9))
        ff, phi = float(rng.uniform(-1e-3, 1e-3)), float(rng.uniform(0, 1))
        w = rng.choice([-1.0, 1.0], nobs).astype(np.float32)
        raw = np.clip(rng.normal(0, 3000, (pt + nobs * ncodes, 2 * nch)), -32768, 32767).astype(np.int16)      # not a frame more than needed
        tag = f"combination {it}: sliding nobs={nobs} ncodes={ncodes} nlag={nlag} nch={nch} ch={ch} pt={pt}"
        got = tracking.sliding_dot(raw, w, nobs, ncodes, nlag, pt=pt, ff=ff, phi=phi, scale=1.0 / 32768.0, n_channels=nch, channel=ch)
        x = raw[:, 2 * ch].astype(np.float64) + 1j * raw[:, 2 * ch + 1]
        for p in range(ncodes):
            i = np.arange(p * nobs, (p + 1) * nobs)
            yy = x[pt + i] / 32768.0 * np.exp(-2j * np.pi * (ff * i + phi))
            ref = orc.sliding_dot(yy, w.astype(np.float64), nlag)
            assert np.abs(got[p] - ref).max() <= 4e-6 * np.abs(ref).max() + 1e-9, (tag, p, float(np.abs(got[p] - ref).max()), float(np.abs(ref).max()))


def test_randomised_tracked_flows():
    """Random captures through the tracked multi-code flow, all three scripts' flavours, against the oracle's restatement: random
    chunk length (a whole number of codes), capture length (incl. a ragged tail of whole codes and of a code fraction), carrier,
    delays that jump once or twice (sample loss: re-alignments, sometimes in the first chunk, sometimes across a chunk end), strong
    to weak signals, fp32 and fp64.  The control flow decides on thresholds, so a window that is measured a hair differently shows
    up as a different `moved` list: everything must agree.  TWX_SWEEP_OPTIONS raises the count."""
    from amaranth_twstft_amd.tracked import TrackedRanging
    rng = np.random.default_rng(2718 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "6"))
    for it in range(ncomb):
        mode, OP = [("ranging", 0), ("lo", 0), ("re", 0), ("re", 1)][int(rng.integers(0, 4))]
        m = orc.tracked_mode(mode, OP)
        car = float(rng.uniform(-3000, 3000)) if mode != "re" else (m["band"][0] + m["band"][1]) / 4 + float(rng.uniform(-500, 500))
        per_chunk = int(rng.integers(8, 41))
        nseg = int(rng.integers(1, 4))
        parts, delay = [], int(rng.integers(100, 15000))
        amp = int(rng.choice([150, 500, 2000])); sigma = float(rng.choice([100.0, 300.0, 800.0]))
        for sgm in range(nseg):
            chips, n, a = _tracked_capture(ncodes=int(rng.integers(per_chunk // 2 + 1, 2 * per_chunk + 5)), df=car, delay=delay, seed=int(rng.integers(1, 10 ** 6)), sigma=sigma, amp=amp)
            parts.append(a)
            delay = (delay + int(rng.choice([-1, 1])) * int(rng.integers(30, 3000))) % n
        raw = np.concatenate(parts)
        raw = raw[: raw.shape[0] - int(rng.choice([0, 0, n // 3, 7]))]                 # sometimes a code fraction at the end
        Lc = per_chunk * n
        precision = str(rng.choice(["f32", "f32", "f64"]))
        tag = f"combination {it}: {mode}/{OP} {precision} chunk {per_chunk} codes, {raw.shape[0] / n:.2f} codes, carrier {car:.1f}, {nseg} segments, amp {amp} sigma {sigma}"
        want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, band=m["band"], carrier=m["carrier"], indice_floor=m["indice_floor"])
        with TrackedRanging(chips, fs=FS, Nint=1, ls_samples=Lc, mode=mode, OP=OP, precision=precision) as tr:
            got = tr.run(raw)
        try:
            assert got["kbon"] == want["kbon"] and got["df"] == want["df"]
            assert got["moved"] == want["moved"] and np.allclose(got["movedval"], want["movedval"])
            assert got["indice1"] == want["indice1"]
            if len(want["xval"]):
                gx, wx = np.abs(np.array(got["xval"])), np.abs(np.array(want["xval"]))
                assert np.abs(gx - wx).max() <= MAG_TOL * wx.max()
                assert np.abs(np.array(got["correction1"]) - np.array(want["correction1"])).max() < 2e-4
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from e


def test_randomised_caf_ranges():
    """Random Doppler ranges of the delay x Doppler surface against the oracle's shift-and-correlate: window length (k_row_caf for
    N2 = 400 / 256, k_rowd_caf with several bins per workgroup for N2 = 4000), precision, first bin anywhere in +-3 N1 and beyond
    (the k1 rotation carries into k2, negative bins), 1 to ~150 bins (whole and ragged groups of bins per workgroup and per
    launch), one / two channels, signal from strong to absent (the arg-max of every bin decided among noise peaks): every bin's lag
    bit-exact, its peak within 1e-6.  TWX_SWEEP_OPTIONS raises the count."""
    rng = np.random.default_rng(55555 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "8"))
    shapes = [(13, 27, 5000), (14, 57, 10000), (16, 45, 32768), (17, 9, 100000)]
    for it in range(ncomb):
        bitlen, taps, nchips = shapes[int(rng.integers(0, len(shapes)))]
        chips = chips_for(bitlen, taps, nchips)
        n = 2 * nchips
        precision = str(rng.choice(["f32", "f32", "f64"]))
        nch = int(rng.integers(1, 3)); ch = int(rng.integers(0, nch))
        nb = int(rng.choice([1, 2, rng.integers(3, 40), rng.integers(40, 150)])) if n <= 65536 else int(rng.choice([1, 7, rng.integers(8, 40)]))
        amp = int(rng.choice([0, 80, 600]))
        true_bin = int(rng.integers(-2000, 2000))
        chans = [synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256, fstep=synth.fstep_for_df(true_bin * FS / n, FS), phi0=int(rng.integers(0, 2 ** 31)), amp=amp,
                                   noise_gain=synth.noise_gain_for_sigma(400.0), seed=int(rng.integers(1, 10 ** 6)), stream=c) for c in range(nch)]
        raw = synth.synth_capture(n, chips, 2, chans)
        with Correlator(chips, fs=FS, Nint=0, precision=precision) as cor:
            n1 = int(cor.info.n1)
            k_lo = int(rng.choice([true_bin - nb // 2, rng.integers(-3 * n1, 3 * n1), -n1 - 1, n1 - nb + 1, rng.integers(-n // 2 + 1, n // 2 - nb)]))
            pk, lag = cor.caf_bins(raw, k_lo, k_lo + nb - 1, n_channels=nch, channel=ch)
        d = orc.deinterleave(raw, nch, ch)
        d = d - d.mean()
        ks, pk_o, lag_o = orc.caf_bins_shift(d, orc.make_fcode(orc.make_code(chips, 2)), k_lo, k_lo + nb - 1)
        tag = f"combination {it}: N={n} (N1={n1}) {precision} nch={nch} ch={ch} bins {k_lo}..{k_lo + nb - 1} amp {amp} true bin {true_bin}"
        assert np.array_equal(lag, lag_o), (tag, np.nonzero(lag != lag_o)[0][:5])
        assert np.abs(pk - pk_o).max() <= (MAG_TOL if precision == "f32" else 1e-12) * pk_o.max(), tag


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_records_do_not_depend_on_the_batch_size(precision):
    """The same 37 two-channel windows through contexts of 1, 3, 8, 16, 37 and 64 windows per launch (whole and ragged last batches,
    one to many batches on the three pipeline slots), carrier searched and supplied, one channel and all channels: the records are
    byte-identical — no result depends on how the windows were grouped (the statistics are exact integer sums, every reduction has a
    fixed order)."""
    chips, raw = _capture(13, 27, 5000, 37, seed=77)
    n = 2 * len(chips)
    band = band_numpy(FS, n)
    dfs = np.linspace(-900.0, 900.0, 37)
    ref = None
    for mb in (1, 3, 8, 16, 37, 64):
        with Correlator(chips, fs=FS, Nint=1, precision=precision, max_batch=mb) as cor:
            assert cor.info.batch == mb
            got = ([(r.indice, r.xval, r.correction, r.df, r.SNRr, r.SNRi, r.puissance, r.puissancecode, r.puissancenoise) for r in cor.process(raw, 2, 1, band=band)],
                   [(r.indice, r.xval, r.correction, r.SNRr, r.puissancenoise) for r in cor.process(raw, 2, 0, df=dfs)],
                   {c: [(r.indice, r.xval, r.df, r.SNRi) for r in v] for c, v in cor.process(raw, 2, -1, band=band).items()})
        if ref is None:
            ref = got
        assert got == ref, mb


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_one_batch_calls_replayed_as_graphs(precision, monkeypatch):
    """TWX_GRAPH=1 (an experiment that stays off: profiles/r05_one_window_graph.txt): a device-resident call of one batch with the
    carrier search on the device is captured once per (input, band, output) and replayed as a hipGraph.  Calls that alternate between two inputs, three bands, two channel selections and two batch sizes —
    each repeated, so every graph is both captured and replayed — return byte for byte what the directly launched chain returns;
    calls the graph path does not take (supplied df, all channels, more than one batch) are unaffected."""
    import torch
    chips, raw = _capture(13, 27, 5000, 8, seed=91)
    n = 2 * len(chips)
    dev_a = torch.from_numpy(raw.reshape(-1)).cuda()
    dev_b = torch.from_numpy(np.ascontiguousarray(raw.reshape(8, n, 4)[::-1]).reshape(-1)).cuda()       # the windows in reverse order
    bands = [band_numpy(FS, n), band_numpy(FS, n, 500.0, 3000.0), band_godual(FS, n)]
    key = lambda r: (r.indice, r.xval, r.correction, r.df, r.SNRr, r.SNRi, r.puissance, r.puissancecode, r.puissancenoise)
    plan = [(d, nw, ch, b) for d in (dev_a, dev_b) for nw in (1, 8) for ch in (0, 1) for b in range(3)]

    def run(graph):
        monkeypatch.setenv("TWX_GRAPH", graph)
        out = []
        with Correlator(chips, fs=FS, Nint=1, precision=precision, max_batch=8) as cor:
            for rep in range(3):
                for d, nw, ch, b in plan:
                    out.append([key(r) for r in cor.process_dev(d.data_ptr(), nw, 2, ch, band=bands[b])])
            out.append([key(r) for r in cor.process_dev(dev_a.data_ptr(), 8, 2, 0, df=np.linspace(-700.0, 700.0, 8))])
            out.append({c: [key(r) for r in v] for c, v in cor.process_dev(dev_a.data_ptr(), 8, 2, -1, band=bands[0]).items()})
        with Correlator(chips, fs=FS, Nint=1, precision=precision, max_batch=3) as cor:                 # 8 windows = three batches: direct
            out.append([key(r) for r in cor.process_dev(dev_b.data_ptr(), 8, 2, 1, band=bands[0])])
        return out

    direct, graphs = run("0"), run("1")
    assert graphs == direct
    assert direct[0] != direct[3] and direct[0] != direct[12]      # the plan's calls do differ (other channel, other input)


def test_fine_frequency_step_with_other_options():
    """TWX_FLAG_FINE_FREQ (the phase-drift fine carrier step of experiments/221219_twoway/processing/godual_ranging.py:26-30; it needs
    fs/3 samples, so N = 2e6) combined with the other options — interpolation factor, variance convention, precision, one- / two-channel
    frames, all-channel call — against orc.processing(fine_freq=True).  The golden test above pins the default combination to the
    reference's own return values; this one walks around it."""
    rng = np.random.default_rng(1123 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    nchips, n = 1_000_000, 2_000_000
    chips = chips_for(21, 5, nchips)
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    freq = orc.freq_axis(FS, n)
    k = orc.band_numpy(freq)
    band = band_numpy(FS, n)
    temps = np.arange(n) / FS
    for it in range(int(os.environ.get("TWX_SWEEP_OPTIONS", "4"))):
        Nint = int(rng.choice([0, 1, 1, 2])); ddof = int(rng.integers(0, 2)); precision = str(rng.choice(["f32", "f64"]))
        nch = int(rng.integers(1, 3)); allc = nch == 2 and bool(rng.integers(0, 2)); ch = int(rng.integers(0, nch))
        chans = [synth.SynthParams(delay_q8=int(rng.integers(0, n)) * 256 + 31, fstep=synth.fstep_for_df(float(rng.uniform(-3000, 3000)) + 2.3 * c, FS),
                                   phi0=int(rng.integers(0, 2 ** 31)), amp=int(rng.choice([300, 1500])), noise_gain=synth.noise_gain_for_sigma(400.0),
                                   seed=int(rng.integers(1, 10 ** 6)), stream=c) for c in range(nch)]
        raw = synth.synth_capture(n, chips, 2, chans)
        with Correlator(chips, fs=FS, Nint=Nint, var_ddof=ddof, precision=precision, fine_freq=True) as cor:
            got = cor.process(raw, nch, -1 if allc else ch, band=band)
        for c in (range(nch) if allc else (ch,)):
            g = got[c][0] if allc else got[0]
            d = orc.deinterleave(raw, nch, c)
            d = d - d.mean()
            o = orc.processing(d, k, freq, temps, fcode, code, Nint=Nint, fs=FS, fine_freq=True, ddof=ddof)
            tag = f"combination {it}: Nint={Nint} ddof={ddof} {precision} nch={nch} channel {c} all={allc}"
            assert g.indice == o["indice"], tag
            assert abs(g.df - o["df"]) <= 1e-5 and abs(g.correction - o["correction"]) <= 2e-4, (tag, g.df, o["df"])
            assert abs(abs(g.xval) - abs(o["xval"])) <= 2e-6 * abs(o["xval"]), tag
            for key in ("SNRr", "SNRi", "puissancecode"):
                assert abs(getattr(g, key) - o[key]) <= 3e-4 * max(o["SNRr"], o["SNRi"], o[key]) + 1e-30, (tag, key)
            assert abs(g.puissance - o["puissance"]) <= 1e-6 * o["puissance"], tag


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("variant", ["claudio", "hamming", "unipolar_zero_mean", "qpsk"])
@pytest.mark.parametrize("bitlen,taps,nchips", [(13, 27, 5000), (17, 9, 100000)])
def test_whole_correlation_map_of_the_variants(bitlen, taps, nchips, variant, precision):
    """Every lag of the map for the replica / convention variants too: the claudio convention (ifft of fcode.*conj(ffty), the mirrored
    conjugate of the godual map), the Hamming-windowed spectrum, the 0/1 zero-mean and the complex QPSK replicas."""
    chips = chips_for(bitlen, taps, nchips)
    n = 2 * nchips
    p = synth.SynthParams(delay_q8=(n // 7 + 3) * 256 + 200, fstep=synth.fstep_for_df(750.0, FS), phi0=7, amp=700,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=nchips + 1)
    raw = synth.synth_channel(n, chips, 2, p)
    x = orc.deinterleave(raw, 1, 0)
    x = x - x.mean()
    y = x * np.exp(-2j * np.pi * 750.0 * np.arange(n) / FS)
    fy = np.fft.fft(y)
    kw, cq = {}, None
    if variant == "claudio":
        fcode = orc.make_fcode(orc.make_code(chips, 2), "claudio")
        kw = dict(convention="claudio")
    elif variant == "hamming":
        fcode = orc.make_fcode(orc.make_code(chips, 2), "hamming")
        kw = dict(window="hamming")
    else:
        cq = chips_for(bitlen, taps + 2, nchips) if variant == "qpsk" else None
        fcode = np.conj(np.fft.fft(orc.make_code_variant(chips, cq, 2, unipolar=True, zero_mean=True)))
        kw = dict(chips_q=cq, code_levels="unipolar", code_zero_mean=True)
    tol = 2e-6 if precision == "f32" else 1e-12
    for Nint in (0, 1):
        with Correlator(chips, fs=FS, Nint=Nint, precision=precision, **kw) as cor:
            z = cor.xcorr_map(raw, 750.0, n_channels=1, channel=0)
        if variant == "claudio":
            r = 2 * Nint + 1
            mul = fcode * np.conj(fy)                                           # claudio_aligned_code_ranging_separate.m:59-61
            pad = np.zeros(r * n, dtype=complex)
            pad[:n // 2] = mul[:n // 2]
            pad[-(n // 2):] = mul[-(n // 2):]
            zc = np.fft.ifft(pad)
            # the map entry always returns the godual form (header); the claudio map is its mirrored conjugate:
            # prnmap_c[m] = conj(prnmap_g[(M - m) mod M]) — which is what k_peak applies to the index and the samples
            zr = np.conj(zc[(-np.arange(r * n)) % (r * n)])
        else:
            zr = orc.xcorr_interp(fy, fcode, Nint)
        err = np.abs(z - zr)
        assert err.max() <= tol * np.abs(zr).max(), (variant, precision, Nint, int(err.argmax()), float(err.max() / np.abs(zr).max()))


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_degenerate_windows(precision):
    """Windows the formulas divide by zero on: all zeros, a constant (the mean removal leaves zeros), one impulse, a single
    step, all in one call beside an ordinary window.  The library and the oracle (numpy, like the reference) agree on the lag
    — ties resolve to the FIRST maximum in both — and on WHICH quantities are not numbers; an ordinary window in the same batch is
    not disturbed."""
    import warnings
    chips = chips_for(13, 27, 5000)
    n = 10000
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    temps = np.arange(n) / FS
    p = synth.SynthParams(delay_q8=777 * 256, fstep=synth.fstep_for_df(0.0, FS), phi0=1, amp=400, noise_gain=synth.noise_gain_for_sigma(300.0), seed=2)
    wins = [np.zeros((n, 2), np.int16), np.full((n, 2), 1234, np.int16), np.zeros((n, 2), np.int16), np.zeros((n, 2), np.int16), synth.synth_channel(n, chips, 2, p).reshape(n, 2)]
    wins[2][4321, 0] = 20000                                           # one impulse
    wins[3][:, 0] = np.where(np.arange(n) < n // 2, 3000, -3000)      # one step, zero mean (a periodic wave would tie its own periods)
    raw = np.concatenate(wins)
    with Correlator(chips, fs=FS, Nint=1, precision=precision) as cor:
        got = cor.process(raw, 1, 0, df=0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            for w, g in enumerate(got):
                d = orc.deinterleave(wins[w], 1, 0)
                d = d - d.mean()
                o = orc.processing(d, None, None, temps, fcode, code, Nint=1, fs=FS, df=0.0)
                assert g.indice == o["indice"], (w, g.indice, o["indice"])
                for key in ("correction", "SNRr", "SNRi", "puissancenoise", "puissancecode", "puissance"):
                    a, b = getattr(g, key), o[key]
                    assert np.isnan(a) == np.isnan(b), (w, key, a, b)
                    if not np.isnan(b) and w >= 2:
                        tol = 2e-4 if key == "correction" else 3e-4 * max(abs(b), abs(o["SNRr"]), abs(o["SNRi"]), 1e-300) + 1e-30
                        assert abs(a - b) <= tol, (w, key, a, b)
                if w >= 2:
                    assert abs(abs(g.xval) - abs(o["xval"])) <= 2e-6 * abs(o["xval"]), w
                else:
                    assert g.xval == 0 and g.puissance == 0


def test_all_channels_from_one_copy(tmp_path):
    """channel = -1: both channels of every window from one upload / one pass over the file equal the per-channel calls
    (host buffer with more chunks than slots, device-resident, file)."""
    chips, raw = _capture(14, 43, 10000, 9, seed=23)
    n = 20000
    band = band_numpy(FS, n)
    dfs = np.stack([np.full(9, 1780.75), np.zeros(9)], axis=1) + np.arange(9)[:, None] * 0.5
    with Correlator(chips, fs=FS, Nint=1, max_batch=2) as cor:
        sep = {c: cor.process(raw, 2, c, band=band) for c in (0, 1)}
        both = cor.process(raw, 2, -1, band=band)
        sep_df = {c: cor.process(raw, 2, c, df=dfs[:, c]) for c in (0, 1)}
        both_df = cor.process(raw, 2, -1, df=dfs)
        path = tmp_path / "1670000001.bin"
        raw.tofile(path)
        both_file = cor.process_file(str(path), 2, -1, band=band)
        rng = cor.ranging(raw, n_channels=2)
    for c in (0, 1):
        for a, b_, f, r in zip(sep[c], both[c], both_file[c], rng[c]):
            for x in (b_, f, r):
                assert (a.indice, a.xval, a.correction, a.df, a.SNRr, a.puissance) == (x.indice, x.xval, x.correction, x.df, x.SNRr, x.puissance)
        for a, b_ in zip(sep_df[c], both_df[c]):
            assert (a.indice, a.xval, a.df) == (b_.indice, b_.xval, b_.df)
    assert len(both[0]) == len(both[1]) == 9


@pytest.mark.parametrize("nobs,ncodes,nlag,pt", [(400000, 24, 28, 5), (40000, 9, 8, 3), (40000, 9, 14, 0), (9004, 3, 8, 1), (40004, 5, 14, 2)])
def test_sliding_dot_on_complex_float_samples(nobs, ncodes, nlag, pt):
    """twx_sliding_dot_cdev: the same correlator on complex-float samples resident on the device (the x2-interpolated stream the
    DLL/PLL receiver tracks on, rxcomplex.cpp:477,602), both kernel forms, against the definition in fp64."""
    import ctypes as C2
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(nobs + nlag)
    w = rng.choice([-1.0, 1.0], nobs).astype(np.float32)
    n = pt + nobs * ncodes + 8
    x = (rng.normal(0, 0.1, n) + 1j * rng.normal(0, 0.1, n)).astype(np.complex64)
    xd = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(-1, 2)).to(dev)
    wd = torch.from_numpy(w).to(dev)
    out = torch.zeros((ncodes, 2 * nlag + 1, 2), dtype=torch.float64, device=dev)
    ff, phi, scale = 3.1e-5, 0.37, 1.4142135624
    with Correlator(lfsr=(14, 43, 10000), fs=FS) as cor:
        L.check(cor._lib.twx_sliding_dot_cdev(cor._h, xd.data_ptr(), n, pt, nobs, ncodes, nlag, wd.data_ptr(), ff, phi, scale, out.data_ptr()), cor._h)
        cor.synchronize()
    got = out.cpu().numpy()
    got = got[..., 0] + 1j * got[..., 1]
    xx = x.astype(np.complex128)
    for p in range(ncodes):
        i = np.arange(p * nobs, (p + 1) * nobs)
        y = scale * xx[pt + i] * np.exp(-2j * np.pi * (ff * i + phi))
        ref = orc.sliding_dot(y, w.astype(np.float64), nlag)
        assert np.abs(got[p] - ref).max() <= 3e-6 * np.abs(ref).max() + 1e-12


def test_clipped_capture_power_is_the_same_on_every_loader_path():
    """A capture that sits on the negative rail: (I, Q) = (-32768, -32768) gives I^2 + Q^2 = 2^31, one more than an int holds.  The
    exact integer sums (k_sums: 16-byte path for an aligned single-channel window, scalar path otherwise, k_sums_deint2 for all
    channels at once) must agree with each other and with the oracle — puissance / puissancenoise of godual_ranging.m:46-48."""
    chips, raw2 = _capture(14, 43, 10000, 2, seed=91)
    n = 2 * len(chips)
    raw2 = raw2.reshape(-1, 4).copy()
    rng = np.random.default_rng(4)
    hit = rng.random(raw2.shape[0]) < 0.3
    raw2[hit, 0] = raw2[hit, 1] = -32768                                        # channel 0 clipped on 30 % of its samples
    raw2[:, 2:] = raw2[:, :2]                                                   # channel 1 = the same signal
    one = raw2[:, :2].copy().reshape(-1)
    band = band_numpy(FS, n)
    ref = orc.ranging(one, chips, fs=FS, Nint=1, n_channels=1, channels=(0,), band="numpy")[0]
    with Correlator(chips, fs=FS, Nint=1) as cor:
        a = cor.process(one, n_channels=1, channel=0, band=band)                                   # 16-byte path
        b = cor.process(raw2.reshape(-1), n_channels=2, channel=0, band=band)                      # scalar path (nch = 2)
        c = cor.process(raw2.reshape(-1), n_channels=2, channel=1, band=band)
        d = cor.process(raw2.reshape(-1), n_channels=2, channel=ALL_CHANNELS, band=band)           # k_sums_deint2
    for w in range(2):
        want = ref[w]
        for got in (a[w], b[w], c[w], d[0][w], d[1][w]):
            assert got.indice == want["indice"]
            assert got.puissance == a[w].puissance and got.puissancenoise == a[w].puissancenoise  # bit-identical across the loader paths
            assert abs(got.puissance - want["puissance"]) <= 1e-6 * want["puissance"]
            assert abs(got.puissancenoise - want["puissancenoise"]) <= 1e-6 * want["puissancenoise"] + 5e-7 * want["puissancecode"]
