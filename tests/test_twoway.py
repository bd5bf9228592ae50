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


def test_session_files_and_1s_writer(tmp_path):
    """go_1s.m:83-139,251-268: the four files of a session found by the script's naming rules, combined, and the
    <MJD>.1s file written with the script's header/row format and file name (Octave num2str of the MJD)."""
    import gzip, io, os
    from scipy.io import savemat
    rng = np.random.default_rng(9)
    n = 25 * 6 + 11
    t = np.arange(n) / 25.0
    sat = 0.26e9 + 5.0 * t
    recs = {"OP/localclaudio1674402311_2.mat.gz": _record(n, 700.0, rng), "OP/remoteclaudio1674402311_1.mat.gz": _record(n, sat + 700.0 + 37.5, rng),
            "LTFB/localclaudio1674402314_1.mat.gz": _record(n, 900.0, rng), "LTFB/remoteclaudio1674402314_2.mat.gz": _record(n, sat + 900.0 - 37.5, rng)}
    for rel, r in recs.items():
        os.makedirs(tmp_path / os.path.dirname(rel), exist_ok=True)
        buf = io.BytesIO()
        savemat(buf, {k: np.asarray(v).reshape(1, -1) for k, v in r.items()} | {"SNR1r": np.ones((1, len(r["xval1"]))), "SNR1i": np.ones((1, len(r["xval1"])))})
        with gzip.open(tmp_path / rel, "wb") as f:
            f.write(buf.getvalue())
    files, ts = twoway.session_files(str(tmp_path), "localclaudio1674402311_2.mat.gz")
    assert ts == 1674402314.0 and files["lt_remote"].endswith("remoteclaudio1674402314_2.mat.gz")
    out = twoway.process_sessions(str(tmp_path), out_dir=str(tmp_path))
    assert len(out) == 1
    mjd, tw, path = out[0]
    # 2023-01-22 15:45:14 UTC = MJD 59966.6564 ; the script adds +0.5-0.084 on top of JD-2400000.5 (:133)
    assert abs(mjd - (59966.0 + (15 + 45 / 60 + 14 / 3600) / 24 + 0.5 - 0.084)) < 1e-9
    assert os.path.basename(path) == twoway.octave_num2str(mjd) + ".1s" and os.path.basename(path).startswith("59967.0")
    lines = open(path).read().split("\n")
    assert lines[0] == "# MJD\t\tOPlocal\tOPremote\tLTFBlocal\tLTBBremote"
    rows = [[float(v) for v in l.split("\t")] for l in lines[1:] if l]
    assert len(rows) == tw.one_second.shape[0] >= 4
    assert abs(rows[1][0] - (mjd + 1 / 86400)) < 1e-6 and abs(rows[0][1] - 700.0) < 0.1 and abs(rows[0][3] - 900.0) < 0.1
    assert abs(0.5 * ((rows[2][2] - rows[2][1]) - (rows[2][4] - rows[2][3])) - 37.5) < 0.1
    assert twoway.octave_num2str(60000.0) == "60000" and twoway.octave_num2str(3.14159265) == "3.1416"
    assert abs(twoway.julian_day(2000, 1, 1.5) - 2451545.0) < 1e-9


def test_cpp_twin_container(tmp_path):
    """processing/CPP/main.cpp:541-647,786-798: <capture>C.mat with correction = indice+corr, SNR in dB, n x 1 columns."""
    from scipy.io import loadmat
    from amaranth_twstft_amd import results_io
    from amaranth_twstft_amd.correlator import WindowResult
    res = [WindowResult(3935295 + i, -0.25 + 0.1 * i, 1 + 2j, 0.5j, 0.25, np.zeros(7, complex), 1780.75, -1, 2e-7, 1e-7, 4e4, 1e-3, 5e3) for i in range(3)]
    assert results_io.cpp_mat_name("/data/1670074501.bin", 0) == "/data/1670074501C.mat"
    assert results_io.cpp_mat_name("1670074501.bin", 1) == "remote1670074501C.mat"
    path = results_io.save_cpp_mat(str(tmp_path / "1670074501.bin"), res, res)
    m = loadmat(path)
    assert m["correction1"].shape == (3, 1) and abs(m["correction1"][1, 0] - (3935296 - 0.15)) < 1e-9
    assert abs(m["SNR2"][0, 0] - 10 * np.log10(3e-7)) < 1e-12 and abs(m["puissance1code"][0, 0] - (-30.0)) < 1e-9
    assert m["xval1m1"][2, 0] == 0.5j and set(k for k in m if not k.startswith("__")) == {
        f"{v}{c}{s}" for c in "12" for v, s in (("correction", ""), ("SNR", ""), ("df", ""), ("puissance", ""), ("puissance", "code"), ("xval", ""), ("xval", "m1"), ("xval", "p1"))}
