"""CPU: result containers keep the reference's schema (godual_ranging.m:96,126-131)."""
import io

import numpy as np

from amaranth_twstft_amd import results_io
from amaranth_twstft_amd.correlator import WindowResult


def _res(i):
    return WindowResult(3935295 + i, -0.005 * i, 1 + 2j, 0.5 + 1j, 0.4 + 0.9j, np.zeros(7, complex), 1780.75, 10, 0.03, 0.02,
                        590000.0, 8000.0, 61000.0)


def test_mat_schema_roundtrip(tmp_path):
    import scipy.io
    r1, r2 = [_res(i) for i in range(4)], [_res(10 + i) for i in range(4)]
    p = tmp_path / "1655300700.mat"
    results_io.save_mat(str(p), r1, r2, code=np.array([1, -1, 1, 1]))
    d = scipy.io.loadmat(str(p))
    for k in ("indice1", "indice2", "correction1", "correction2", "df1", "df2", "SNR1r", "SNR1i", "SNR2r", "SNR2i",
              "puissance1", "puissance1code", "puissance1noise", "xval1", "xval1m1", "xval1p1", "code"):
        assert k in d, k
    assert d["indice1"].shape == (1, 4) and d["indice1"][0, 0] == 3935296.0       # 1-based like Octave's max()
    assert d["xval2"].dtype == np.complex128 and d["xval2"][0, 1] == 1 + 2j


def test_tsv_row_format():
    rows = list(results_io.tsv_rows([_res(0)], [_res(1)], 5e6, 1))
    assert rows[0].startswith("n\tdt1\tdf1")
    f = rows[1].rstrip("\r\n").split("\t")
    assert f[0] == "1" and f[1] == "%.12f" % (3935295 / 5e6 / 3) and f[2] == "1780.750" and len(f) == 9


def test_tracked_mat_roundtrip_feeds_twoway(tmp_path):
    """tracked .mat variable set (claudio_aligned_code_ranging_separate.m:207) → loadmat → twoway (go_1s.m:83-95)."""
    from scipy.io import loadmat
    from amaranth_twstft_amd import results_io, twoway
    n = 60
    out = dict(xval=[(1000 + i) * np.exp(0.2j) for i in range(n)], indice1=[21.0 + (i % 2) for i in range(n)],
               correction1=[0.1] * n, SNR1r=[1e-3] * n, SNR1i=[2e-3] * n, puissance1=[5.0] * n, df=[12.5, 12.0],
               moved=[1], movedval=[777.0])
    p = tmp_path / "rangingclaudio_x.mat"
    results_io.save_tracked_mat(str(p), out, code=[1, -1, 1])
    m = loadmat(str(p))
    assert m["xval1"].shape == (1, n) and np.iscomplexobj(m["xval1"]) and m["df"].shape == (1, 2)
    assert m["moved"].ravel().tolist() == [1] and m["code"].ravel().tolist() == [1, -1, 1]
    k, trunc = twoway.valid_codes(m["xval1"].ravel())
    d = twoway.delays_ns(m["indice1"].ravel(), m["correction1"].ravel(), k)
    assert not trunc and len(d) == n - 11 and abs(d[0] - (21.0 + 0.1 / 3) / 5e6 * 1e9) < 1e-9


def test_mat_writers_follow_the_schema_of_the_reference_archives(tmp_path):
    """tests/golden/mat_schema.json = names/shapes/dtypes of result files kept in the reference repository
    (tools/make_golden.py:gen_mat_schema).  Every variable of the 2023 tracked-script archive must come out of
    save_tracked_mat, every variable of the 2022 two-channel archive that the current script still saves must come out of
    save_mat, as 1 x n rows of the same dtype kind."""
    from scipy.io import loadmat
    from amaranth_twstft_amd import results_io
    from amaranth_twstft_amd.correlator import WindowResult
    from tests.helpers import load_golden
    schema = load_golden("mat_schema.json")["files"]
    n = 7
    out = dict(xval=[1 + 1j] * n, indice1=[21.0] * n, correction1=[0.1] * n, SNR1r=[1e-3] * n, SNR1i=[2e-3] * n,
               puissance1=[5.0] * n, df=[12.5], moved=[1], movedval=[7.0], puissancecode=3.0, puissancenoise=4.0)
    p = tmp_path / "t.mat"
    results_io.save_tracked_mat(str(p), out, code=np.ones(10))
    got = loadmat(str(p))
    ref = schema["experiments/230315_analysis_100k/local1674402311.mat.gz"]
    for name, d in ref.items():
        assert name in got, name
        assert got[name].ndim == 2 and got[name].shape[0] == 1 and got[name].dtype.kind == np.dtype(d["dtype"]).kind, name
        if d["shape"] == [1, 1]:
            assert got[name].shape == (1, 1), name
    res = [WindowResult(10, 0.1, 1 + 1j, 1j, 1, np.zeros(7, complex), 5.0, 3, 1e-3, 1e-3, 2.0, 1.0, 1.0) for _ in range(n)]
    q = tmp_path / "g.mat"
    results_io.save_mat(str(q), res, res, code=np.ones(10))
    got = loadmat(str(q))
    ref = schema["experiments/220616_Besancon/1655300700.mat.gz"]
    for name, d in ref.items():
        if name == "df":                      # the 2022 variant saved one df; godual_ranging.m:126-131 saves df1 df2
            assert "df1" in got and "df2" in got
            continue
        assert name in got and got[name].shape == (1, n) and got[name].dtype.kind == np.dtype(d["dtype"]).kind, name
