"""CPU: the device FFT templates (csrc/twx_fft.h) run under a thread-loop emulation with g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.slow
def test_fft_templates_on_cpu(tmp_path):
    exe = tmp_path / "fft_emul"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "cpu", "fft_emul.cpp")],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "ALL OK" in out.stdout and "FAIL" not in out.stdout


@pytest.mark.slow
def test_rowd_transform_on_cpu(tmp_path):
    """The DIF/DIT row transform (RowD: one all-to-all stage, two wave-local stages) against a direct DFT."""
    exe = tmp_path / "rowd_emul"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "cpu", "rowd_emul.cpp")],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "ALL OK" in out.stdout and "FAIL" not in out.stdout
