"""CPU: two-way combination (acquisition/go_1s.m:83-268) — known-answer checks on synthetic series."""
import numpy as np

from amaranth_twstft_amd import twoway


def _record(n, delay_ns, rng, fs=5e6, lead=3, amp=1000.0, jitter=0.05):
    """A tracked-correlator record: `lead` empty codes, then n codes at `delay_ns` (array or scalar)."""
    d = np.broadcast_to(np.asarray(delay_ns, dtype=float), (n,)) + rng.normal(0, jitter, n)
    samples = d * 1e-9 * fs
    ind = np.floor(samples)
    corr = (samples - ind) * 3
    xval = np.concatenate((np.full(lead, amp / 50), np.full(n, amp))) * np.exp(1j * 0.3)
    return dict(xval1=xval, indice1=np.concatenate((np.zeros(lead), ind)), correction1=np.concatenate((np.zeros(lead), corr)))


def test_valid_codes_and_sample_loss():
    x = np.array([1] * 3 + [100] * 40 + [1] * 2 + [100] * 5, dtype=float)
    k, trunc = twoway.valid_codes(x)
    assert trunc and k[0] == 13 and k[-1] == 42                 # first 10 dropped, cut at the gap
    k, trunc = twoway.valid_codes(np.array([1] * 3 + [100] * 40, dtype=float))
    assert not trunc and k[0] == 13 and k[-1] == 41             # …and the last one dropped
    lo = np.array([5.0, 5.1, 5.0, 9.0, 9.1])
    cut, at = twoway.cut_at_sample_loss(lo)
    assert at == 3 and np.array_equal(cut, lo[:2])


def test_two_way_difference_and_one_second_rows():
    rng = np.random.default_rng(5)
    n = 25 * 8 + 11
    t = np.arange(n) / 25.0
    sat = 0.26e9 + 5.0 * t                                       # common satellite path, 5 ns/s drift
    op_lo, lt_lo = 700.0, 900.0
    clock = 37.5                                                 # (OP-LTFB)/2 observable, ns
    op_re = sat + op_lo + clock
    lt_re = sat + lt_lo - clock
    rec = dict(op_local=_record(n, op_lo, rng), op_remote=_record(n, op_re, rng),
               lt_local=_record(n, lt_lo, rng), lt_remote=_record(n, lt_re, rng))
    tw = twoway.session(**rec)
    assert len(tw.res) == n - 11                                 # 10 leading codes + the last one dropped
    assert abs(tw.resmean - clock) < 0.02 and tw.resstd < 0.2
    assert abs(tw.resmean25 - clock) < 0.02 and tw.resstd25 < tw.resstd
    assert abs(tw.opslope[0] - 5.0) < 0.01 and abs(tw.ltslope[0] - 5.0) < 0.01
    assert np.nanmax(np.abs(tw.res2 - clock)) < 0.5
    assert tw.one_second.shape == ((n - 11 - 25 - 1) // 25 + 1, 5)
    row = tw.one_second[2]
    assert row[0] == 2 and abs(row[1] - op_lo) < 0.05 and abs(row[3] - lt_lo) < 0.05
    assert abs(0.5 * ((row[2] - row[1]) - (row[4] - row[3])) - clock) < 0.1


def test_outliers_become_nan():
    oplo = np.full(100, 700.0); ltlo = np.full(100, 900.0)
    opre = np.full(100, 1e6); ltre = np.full(100, 1e6 - 50)
    opre[40] += 30.0
    tw = twoway.combine(oplo, opre, ltlo, ltre)
    assert np.isnan(tw.res[40]) and np.count_nonzero(np.isnan(tw.res)) == 1
    assert abs(tw.resmean - 125.0) < 1e-9
