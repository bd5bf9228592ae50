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
