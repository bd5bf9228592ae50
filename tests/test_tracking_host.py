"""CPU: host arithmetic of the tracking epoch (experiments/231001_DLL_PLL/rxcomplex.cpp:620-745), known answers."""
import numpy as np

from amaranth_twstft_amd import tracking


def _matrices(bps, nlag, true_lag, f_res, phi0, duration, pt, fs, outlier=None):
    """Triangular correlation peak at fractional lag `true_lag`; phase ramps with the residual carrier f_res."""
    lags = np.arange(-nlag, nlag + 1)
    cor = np.zeros((bps - 1, 2 * nlag + 1))
    phi = np.zeros_like(cor)
    for p in range(bps - 1):
        tl = true_lag + 0.01 * np.sin(1.7 * p) + (outlier[1] if outlier and p == outlier[0] else 0.0)   # jitter: IQR > 0
        amp = np.clip(1.0 - np.abs(lags - tl) / 2.0, 0.0, None)          # 2-sample-wide triangle (1 chip at 2 sps)
        cor[p] = amp ** 2
        ph = phi0 + f_res * (p * duration + pt / fs)
        ph += 0.5 * (p % 2)                                              # BPSK data flips: half-cycle jumps
        phi[p] = (ph + 0.5) % 1.0 - 0.5
    return cor, phi


def test_weighted_linear_fit_matches_polyfit():
    rng = np.random.default_rng(0)
    x = np.arange(20.0); y = 3.0 - 0.25 * x + rng.normal(0, 0.01, 20); w = np.ones(20); w[5] = 0
    c0, c1, chi = tracking._wlinear(x, w, y)
    ref = np.polyfit(np.delete(x, 5), np.delete(y, 5), 1)
    assert abs(c1 - ref[0]) < 1e-12 and abs(c0 - ref[1]) < 1e-12 and chi > 0


def test_tracking_epoch_recovers_carrier_and_code_phase():
    fs, nobs, bps, nlag = 5e6, 20000, 25, 28
    duration = nobs / fs
    pt = 1000
    st = dict(fc=1000.0, pt=pt, last_phi=0.1, fs=fs, duration=duration, psbb=1.0)
    cor, phi = _matrices(bps, nlag, true_lag=3.3, f_res=7.4, phi0=0.12, duration=duration, pt=pt, fs=fs, outlier=(9, 6.0))
    out = tracking.tracking_update(cor, phi, nlag, st)
    assert out is not None and out["cnt"] == bps - 2                     # the 6-sample outlier is rejected
    assert abs(out["freq"] - 1007.4) < 1e-6                              # fc + round(c1) + fractional part
    assert st["fc"] == 1007.0 and abs(st["df"] - 0.4) < 1e-6
    assert abs(out["dg"]) < 20.0 and out["sdgd"] < 5.0                    # constant delay up to the 2-ns jitter
    hrc = out["gd"] * fs / 1e9 - pt                                      # HRC estimate in samples relative to pt
    assert abs(hrc - 3.3) < 0.5                                          # the HRC discriminator is biased on a triangle; sign and size right
    assert st["pt"] == int(round((out["gd"]) * fs / 1e9))


def test_tracking_epoch_needs_half_of_the_periods():
    fs, nobs, bps, nlag = 5e6, 20000, 25, 28
    cor = np.zeros((bps - 1, 2 * nlag + 1)); phi = np.zeros_like(cor)
    cor[:, 0] = 1.0                                                      # peak at the window edge everywhere → unusable
    st = dict(fc=0.0, pt=0, last_phi=0.0, fs=fs, duration=nobs / fs)
    assert tracking.tracking_update(cor, phi, nlag, st) is None and st["fc"] == 0.0


def test_prn_sampling_follows_the_reference_formula():
    """rxcomplex.cpp:965-978 evaluated in the same double arithmetic (scalar loop), including its rounding: i/fs*rc
    lands just below an integer for some i, so the replica is the x2 sample-and-hold of the chips
    (godual_ranging.m:64 ``repelems``) except at those isolated samples."""
    import math
    rng = np.random.default_rng(3)
    code = rng.integers(0, 2, 1000) * 2 - 1
    fs, rc, clen = 5e6, 2.5e6, 1000
    for delay in (0.0, 200.0, -100.0, 123.456):
        got = tracking.prn_sampling(2000, code, rc, fs, delay)
        want = np.empty(2000, dtype=np.float32)
        for i in range(2000):
            idx = int(math.floor(math.fmod((float(i) / fs - delay * 1.0e-9) * rc, float(clen))))
            if idx < 0:
                idx += clen
            elif idx >= clen:
                idx -= clen
            want[i] = code[idx]
        assert np.array_equal(got, want)
    r0 = tracking.prn_sampling(2000, code, rc, fs, 0.0)
    assert np.mean(r0 == np.repeat(code, 2)) > 0.97
    r2 = tracking.prn_sampling(3000, code, rc, 7.5e6, 0.0)                  # 3 samples per chip
    assert np.mean(r2 == np.repeat(code, 3)) > 0.97


def test_track_update_equals_the_oracle_restatement_of_the_epoch():
    """twx_track_update (C++, behind the C ABI; tracking.tracking_update is its binding) against oracle.rx_track_epoch, the
    function-by-function restatement of rxcomplex.cpp:593-745, fed with the SAME correlation matrices (computed by the oracle's
    downconv_trk / PRN_mapping / dgemm restatement): every printed quantity and the updated channel state."""
    from oracle import twstft_oracle as orc
    rng = np.random.default_rng(1)
    fs, nobs, bps, nlag = 10e6, 4000, 25, 28
    code = rng.integers(0, 2, 1000) * 2 - 1
    wav = orc.rx_prn_sampling(nobs, code, 2.5e6, fs, 1000).real
    n = nobs * (bps + 2)
    for delay, fres, flips in ((17, 3.4, False), (40, -7.8, True)):
        x = np.roll(np.tile(wav, bps + 2), delay) * np.exp(2j * np.pi * ((1000.0 + fres) / fs * np.arange(n) + 0.1)) * 0.2
        if flips:                                                          # BPSK data: half-cycle phase jumps between code periods
            x = x * np.repeat(rng.integers(0, 2, bps + 2) * 2 - 1, nobs)
        x = x + rng.normal(0, 0.3, n) + 1j * rng.normal(0, 0.3, n)
        st = dict(fc=1000.0, pt=delay - 2, last_phi=0.0, psbb=0.7, duration=nobs / fs, fs=fs)
        st2 = dict(st)
        want = orc.rx_track_epoch(x, wav.astype(complex), st, nobs, bps, nlag, fs)
        pt = st2["pt"]
        obs = orc.rx_downconv_trk(nobs * (bps - 1), nobs, st2["fc"] / fs, float(np.fmod(pt * st2["fc"] / fs, 1.0)), x[pt:])
        res = (obs @ orc.rx_prn_mapping(nobs, nlag, wav.astype(complex)).T) / nobs
        got = tracking.tracking_update(res.real ** 2 + res.imag ** 2, np.arctan2(res.imag, res.real) / 2 / np.pi, nlag, st2, nobs=nobs)
        assert want is not None and got is not None and got["cnt"] == want["cnt"] >= bps - 3
        # the records rx.cpp keeps for MAI_up (:664-666,752-757): peak lags exact, amplitudes and recovered phases to rounding
        assert np.array_equal(got["mai"]["pk_idx"], st["mai"]["pk_idx"]) and np.count_nonzero(got["mai"]["amp"]) >= bps - 3
        assert np.allclose(got["mai"]["amp"], st["mai"]["amp"], rtol=1e-12, atol=0) and np.allclose(got["mai"]["phase"], st["mai"]["phase"], rtol=0, atol=1e-9)
        assert got["mai"]["phase"][-1] == 0.0 and got["mai"]["amp"][-1] == 0.0
        for k in ("freq", "phi", "gd", "dg", "sdgd", "pk"):
            assert abs(got[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), k
        assert (st2["pt"], st2["fc"], st2["pt_prev"]) == (st["pt"], st["fc"], st["pt_prev"]) and abs(st2["last_phi"] - st["last_phi"]) < 1e-12
        assert abs(want["freq"] - (1000.0 + fres)) < 0.5 and st["pt"] == delay


def test_oracle_octave_xcorr_and_epl_step():
    """oracle.octave_xcorr (the LINEAR cross-correlation of gotracking_inv2.m:161-163 with MAXLAG = N) against its definition, both
    evaluation routes; oracle.epl_step: one block of a clean delayed code gives the prompt peak at lag -delay, late / early one sample
    either side, a zero phase discriminator for a carrier-free block."""
    from oracle import twstft_oracle as orc
    from amaranth_twstft_amd import epl, prn
    rng = np.random.default_rng(0)
    for n in (50, 5000):
        a = rng.choice([-1.0, 1.0], n)
        b = rng.normal(size=n) + 1j * rng.normal(size=n)
        z = orc.octave_xcorr(a, b)
        assert z.size == 2 * n + 1 and z[0] == 0 and z[-1] == 0
        for k in (-(n - 1), -7, 0, 3, n - 1):
            ref = sum(a[i + k] * np.conj(b[i]) for i in range(n) if 0 <= i + k < n)
            assert abs(z[k + n] - ref) <= 1e-10 * max(1.0, abs(ref))
    chips = prn.lfsr_chips(13, 27, 2500)
    al, ap, ae = epl.replicas(chips, 2)
    assert np.array_equal(al[1:], ap[:-1]) and np.array_equal(ae[:-1], ap[1:]) and al[0] == ap[-1] and ae[-1] == ap[0]
    x = 100.0 * np.roll(ap, 5)
    st = dict(l=1, doppler_freq=[0.0], time_end=0.0, code_phase=0.0, carrier_phase=0.0)
    o = orc.epl_step(st, x, al, ap, ae)
    assert o["bbp"] == ap.size + 1 - 5 and o["bbl"] == o["bbp"] + 1 and o["bbe"] == o["bbp"] - 1
    assert abs(o["delta_theta"]) < 1e-12 and st["l"] == 2 and len(st["doppler_freq"]) == 2 and abs(st["time_end"] - ap.size / 5e6) < 1e-15
