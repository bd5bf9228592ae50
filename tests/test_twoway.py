"""CPU: two-way combination (acquisition/go_1s.m:83-268) — known-answer checks on synthetic series."""
import numpy as np
import pytest

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
    as_script = twoway.session(**rec)                            # default = go_1s.m as written: the shift of :208-211 hits every code
    tw = twoway.session(**rec, unwrap=False)                     # the plain two-way difference
    assert abs((as_script.resmean - tw.resmean) - 200 / 3) < 1e-9 and abs(as_script.resstd - tw.resstd) < 1e-9
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
    tw = twoway.combine(oplo, opre, ltlo, ltre, unwrap=False)
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


def _tracked_like_record(rng, n, base_ns, drift_ns_s=0.0, lead=4, gap_at=None, loss_at=None, weak_tail=0):
    """What the tracked correlator writes: xval1/indice1/correction1/SNR1r/SNR1i rows, with optional faults — a run of weak
    codes in the middle (gap), a delay jump (sample loss), weak codes at the end."""
    t = np.arange(n) / 25.0
    d = base_ns + drift_ns_s * t + rng.normal(0, 0.05, n)
    if loss_at is not None:
        d[loss_at:] += 200.0                                      # one sample at 5 Msps
    s = d * 1e-9 * 5e6
    ind = np.floor(s)
    amp = np.full(n, 1000.0)
    amp[:lead] = 20.0
    if gap_at is not None:
        amp[gap_at:gap_at + 3] = 15.0
    if weak_tail:
        amp[-weak_tail:] = 10.0
    return dict(xval1=amp * np.exp(1j * rng.uniform(0, 6.28, n)), indice1=ind, correction1=(s - ind) * 3,
                SNR1r=np.full(n, 2e-3) + rng.uniform(0, 1e-4, n), SNR1i=np.full(n, 1e-3))


@pytest.mark.parametrize("case", ["clean", "op_gap", "op_loss", "re_gap", "lt_gap", "lt_re_gap", "lt_re_weak_tail", "lt_short", "too_short"])
def test_session_matches_the_oracle_restatement_of_go_1s(case):
    """twoway.session (product, host arithmetic on the device's records) against oracle.go_1s_session, the line-by-line
    restatement of acquisition/go_1s.m:77-268, over the fault cases the script handles: gaps (:81-84,110-118,140-143,
    160-165), sample loss in the loop-back (:94-101), remote records weaker at the end (:166-170), unequal lengths
    (:176-182), sessions with too few codes (:102)."""
    from oracle import twstft_oracle as orc
    rng = np.random.default_rng(hash(case) % 1000)
    n = 25 * 9 + 17
    kw = dict(op_lo={}, op_re={}, lt_lo={}, lt_re={})
    if case == "op_gap": kw["op_lo"] = dict(gap_at=180)
    if case == "op_loss": kw["op_lo"] = dict(loss_at=170)
    if case == "re_gap": kw["op_re"] = dict(gap_at=150)
    if case == "lt_gap": kw["lt_lo"] = dict(gap_at=200)
    if case == "lt_re_gap": kw["lt_re"] = dict(gap_at=190)
    if case == "lt_re_weak_tail": kw["lt_re"] = dict(weak_tail=30)
    n_lt = n - 40 if case == "lt_short" else n
    n_op = 110 if case == "too_short" else n
    recs = dict(op_lo=_tracked_like_record(rng, n_op, 700.0, **kw["op_lo"]), op_re=_tracked_like_record(rng, n_op, 0.26e9 + 737.5, 5.0, **kw["op_re"]),
                lt_lo=_tracked_like_record(rng, n_lt, 900.0, **kw["lt_lo"]), lt_re=_tracked_like_record(rng, n_lt, 0.26e9 + 862.5, 5.0, **kw["lt_re"]))
    want = orc.go_1s_session(recs["op_lo"], recs["op_re"], recs["lt_lo"], recs["lt_re"])
    got = twoway.session(recs["op_lo"], recs["op_re"], recs["lt_lo"], recs["lt_re"], unwrap=True)
    if case == "too_short":
        assert want is None and got is None
        return
    for name in ("oplo", "opre", "ltlo", "ltre"):
        assert np.array_equal(getattr(got, name), want[name]), name
    assert len(want["oplo"]) > 100 and (case == "clean") == (len(want["oplo"]) == n - 15)    # 4 weak leading codes, 10 dropped, the last one dropped
    assert np.allclose(got.res, want["res"], equal_nan=True, atol=1e-9) and np.allclose(got.res2, want["res2"], equal_nan=True, atol=1e-6)
    assert abs(np.nanmean(want["res"]) - (37.5 + 200 / 3)) < 0.1             # the shift of :210-211 hits every code
    assert np.allclose(got.one_second, want["rows"], atol=1e-9) and want["rows"].shape[0] >= 3
    for name in ("resmean", "resstd", "resmean25", "resstd25"):
        assert abs(getattr(got, name) - want[name]) < 1e-9, name
    assert np.allclose(got.opslope, want["opslope"]) and np.allclose(got.ltslope, want["ltslope"])
    assert abs(twoway.snr_db(recs["op_re"]["SNR1r"], recs["op_re"]["SNR1i"], slice(None)) - want["snrop"]) < 0.2


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


def test_c_abi_cmat_writer_equals_the_python_container(tmp_path):
    """twx_write_cmat (csrc/twx_filedf.hip: GoRanging::save, processing/CPP/main.cpp:521-656, as a C entry point — host only, no GPU
    involved): a MAT-v5 file scipy reads back with the variable set, order, shapes and values of results_io.save_cpp_mat."""
    import ctypes as C
    from scipy.io import loadmat
    from amaranth_twstft_amd import _lib as L, results_io
    from amaranth_twstft_amd.correlator import WindowResult
    lib = L.load()
    rng = np.random.default_rng(3)
    n = 7
    recs = [(L.twx_result * n)(), (L.twx_result * n)()]
    wres = [[], []]
    for c in range(2):
        for i in range(n):
            r = recs[c][i]
            r.indice0 = int(rng.integers(0, 15_000_000)); r.correction = float(rng.uniform(-0.5, 0.5)); r.df = float(rng.normal(0, 2000))
            r.SNRr, r.SNRi = float(rng.uniform(1e-8, 1e-3)), float(rng.uniform(1e-8, 1e-3))
            r.puissance, r.puissancecode, r.puissancenoise = float(rng.uniform(1, 1e6)), float(rng.uniform(1e-6, 10)), 1.0
            for k, name in enumerate(("xval", "xvalm1", "xvalp1")):
                z = getattr(r, name); z[0], z[1] = float(rng.normal()), float(rng.normal())
            wres[c].append(WindowResult(int(r.indice0), r.correction, complex(*r.xval), complex(*r.xvalm1), complex(*r.xvalp1), np.zeros(7, complex), r.df, -1,
                                        r.SNRr, r.SNRi, r.puissance, r.puissancecode, r.puissancenoise))
    for two in (True, False):
        path = tmp_path / ("both.mat" if two else "one.mat")
        assert lib.twx_write_cmat(str(path).encode(), C.cast(recs[0], C.c_void_p), C.cast(recs[1], C.c_void_p) if two else None, n) == 0
        got = loadmat(str(path))
        ref_path = results_io.save_cpp_mat(str(tmp_path / ("r2.bin" if two else "r1.bin")), wres[0], wres[1] if two else None)
        want = loadmat(ref_path)
        keys = [k for k in want if not k.startswith("__")]
        assert [k for k in got if not k.startswith("__")] == keys and len(keys) == (16 if two else 8)
        for k in keys:
            assert got[k].shape == want[k].shape == (n, 1) and got[k].dtype == want[k].dtype, k
            if k.startswith("SNR") or k.endswith("code"):          # 10*log10 in libm here, in numpy there: an ulp apart
                assert np.abs(got[k] - want[k]).max() <= 1e-13 * np.abs(want[k]).max(), k
            else:
                assert np.array_equal(got[k], want[k]), k
    assert lib.twx_write_cmat(str(tmp_path / "nodir" / "x.mat").encode(), C.cast(recs[0], C.c_void_p), None, n) != 0
    assert b"cannot create" in lib.twx_file_df_last_error()
