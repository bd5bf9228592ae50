"""Host logic of the wideband session (no GPU): the four-correlation plan of godual_ranging.m:83-89 / go_1s.m:88,120,147,171."""
import os
import subprocess

import numpy as np
import pytest

from amaranth_twstft_amd.correlator import band_godual, freq_axis
from amaranth_twstft_amd.wideband import godual_plan

FS, N = 5e6, 5_000_000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amaranth_twstft_amd", "csrc")


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


CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANGXX), reason="ROCm clang++ (for _Float16 on the host) is not installed")
def test_fir_matrix_core_table_arithmetic_on_the_cpu(tmp_path):
    """csrc/twx_fir_table.h (geometry + A fragments of k_fir_mfma) through a CPU emulation of the kernel's arithmetic — fp16 (xh, xl) sample pairs
    against (256 h, h) tap pairs in the kernel's Toeplitz layout, float accumulation per wave, the eight partial sums added in wave order —
    against the fp64 direct sum, inside the gate of the GPU tests (2e-6 of the maximum + 1e-3); under UBSan.  Shapes: configs[4], few taps per
    phase, one pair per wave with idle waves, five steps per phase, one phase; a geometry that does not fit is refused (exit 3)."""
    exe = tmp_path / "fir_table_emul"
    subprocess.run([CLANGXX, "-O2", "-std=c++17", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-I" + CSRC, "-o", str(exe),
                    os.path.join(ROOT, "tests", "cpu", "fir_table_emul.cpp")], check=True)
    for ntaps, dec, seed in ((421, 14, 1), (64, 16, 2), (33, 3, 3), (171, 3, 4), (232, 8, 5), (31, 1, 6), (97, 4, 7)):
        r = subprocess.run([str(exe), str(ntaps), str(dec), str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, (ntaps, dec, r.stdout, r.stderr)
        assert "max |err|" in r.stdout
    r = subprocess.run([str(exe), "700", "16", "8"], capture_output=True, text=True)
    assert r.returncode == 3 and "does not fit" in r.stdout
