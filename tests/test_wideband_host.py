"""Host logic of the wideband session (no GPU): the four-correlation plan of godual_ranging.m:83-89 / go_1s.m:88,120,147,171."""
import numpy as np

from amaranth_twstft_amd.correlator import band_godual, freq_axis
from amaranth_twstft_amd.wideband import godual_plan

FS, N = 5e6, 5_000_000


def test_godual_plan_names_codes_and_bands():
    plan = godual_plan(("OP", "LTFB"), FS, N)
    assert list(plan) == ["OPlo", "OPre", "LTFBlo", "LTFBre"]           # go_1s.m's oplo, opre, ltlo, ltre
    assert [v[:2] for v in plan.values()] == [("OP", "OP"), ("OP", "LTFB"), ("LTFB", "LTFB"), ("LTFB", "OP")]
    f = freq_axis(FS, N)
    lo = plan["OPlo"][2]
    assert plan["LTFBlo"][2] == lo == band_godual(FS, N)
    assert -20000 < f[lo[0]] and f[lo[1]] < 20000 and f[lo[0] - 1] <= -20000 and f[lo[1] + 1] >= 20000
    # the remote band of the squared spectrum: +(80..120) kHz in the OP station's capture, -(120..80) kHz in the other one
    a, b = plan["OPre"][2], plan["LTFBre"][2]
    assert 80000 < f[a[0]] < f[a[1]] < 120000 and -120000 < f[b[0]] < f[b[1]] < -80000
    assert a[1] - a[0] == b[1] - b[0]
    assert np.isclose(f[a[0]], -f[b[1]], atol=2.0)
