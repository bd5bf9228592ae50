"""CPU: sanitizer builds of the host-side C++ (SURVEY.md §5 "race detection / sanitizers"; sanitizers run on the CPU build only).

* ``-fsanitize=address,undefined``: the device FFT templates under the thread-loop emulation (fft_emul, rowd_emul), the tracked
  control flow (twx_tracked_core.h, driven by the oracle through the same test functions as tests/test_tracked_host.py, in a
  child interpreter with libasan preloaded), the tracking epoch's host arithmetic (twx_track_core.h, fuzzed), both MEX gateways on their error paths (no GPU here: mexFunction must fail with
  a MEX error, not with a sanitizer report).
* ``-fsanitize=thread``: the library's host threads (twx_workers.h: the worker pool of twx_multi, the piece-wise chunk reader and
  the slot rotation of the ingest pipeline).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amaranth_twstft_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _no_report(r):
    text = r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in text and "runtime error:" not in text and "LeakSanitizer" not in text and "ThreadSanitizer" not in text, text[-4000:]


@pytest.mark.parametrize("src", ["fft_emul", "rowd_emul"])
def test_fft_templates_under_asan_ubsan(tmp_path, src):
    exe = tmp_path / src
    subprocess.run(["g++", "-O1", "-std=c++17", *SAN, "-o", str(exe), os.path.join(ROOT, "tests", "cpu", src + ".cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV, timeout=900)
    _no_report(r)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_tracking_epoch_arithmetic_under_asan_ubsan(tmp_path):
    """csrc/twx_track_core.h (twx_track_update[_mai]: peak pick, high-resolution correlator, order statistics, phase unwrap, the two
    weighted fits, the interference-cancellation records) on 20 000 random correlation matrices incl. edge peaks, flat tops, zeros,
    NaN / infinity entries: no report, invariants hold (tests/cpu/track_fuzz.cpp)."""
    exe = tmp_path / "track_fuzz"
    subprocess.run(["g++", "-O1", "-std=c++17", *SAN, "-I" + CSRC, "-o", str(exe), os.path.join(ROOT, "tests", "cpu", "track_fuzz.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV, timeout=900)
    _no_report(r)
    assert r.returncode == 0 and "track ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_affinity_helpers_under_asan_ubsan(tmp_path):
    """csrc/twx_affinity.h (NUMA node / local CPUs of a GPU from sysfs, binding a worker thread) against a fake sysfs tree."""
    for bus, node, cpus in (("0000:c1:00.0", "1", "0-1"), ("0000:05:00.0", "-1", "0"), ("0000:06:00.0", "0", "zero-three")):
        d = tmp_path / "sys" / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node + "\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    exe = tmp_path / "affinity_test"
    subprocess.run(["g++", "-O1", "-std=c++17", *SAN, "-I" + CSRC, "-o", str(exe), os.path.join(ROOT, "tests", "cpu", "affinity_test.cpp"), "-lpthread"], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "sys")], capture_output=True, text=True, env=ENV, timeout=300)
    _no_report(r)
    assert r.returncode == 0 and "affinity ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_host_threads_under_tsan(tmp_path):
    exe = tmp_path / "threads_tsan"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + CSRC, os.path.join(ROOT, "tests", "cpu", "threads_tsan.cpp"),
                    "-o", str(exe), "-lpthread"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), timeout=900)
    _no_report(r)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_tracked_control_flow_under_asan_ubsan():
    """The SAME tests as tests/test_tracked_host.py (the oracle answers the backend calls) with the control flow compiled with
    ASan + UBSan: a child interpreter with libasan preloaded loads the instrumented tracked_emul.so."""
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(ENV, LD_PRELOAD=libasan, TWX_EMUL_SANITIZE="1", PYTHONPATH=ROOT, ASAN_OPTIONS="detect_leaks=0")   # CPython itself is not leak-clean
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_tracked_host.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "helpers or control_flow or skip_and_known or no_carrier or edge_paths"],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    _no_report(r)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("harness,gateway,args", [
    ("mex_harness", "twstft_processing_mex", ["raw", "{cap}", "{chips}", "{out}", "2", "0", "1", "100", "5e6", "1"]),
    ("mex_harness", "twstft_processing_mex", ["file", "{cap}", "{chips}", "{out}", "2", "1", "df", "12.5", "5e6", "1", "claudio", "ngpu=3", "skip=5", "max=1"]),
    ("mex_tracked_harness", "twstft_tracked_mex", ["{cap}", "{chips}", "{out}", "lo", "0", "5e6", "1"]),
])
def test_mex_gateways_error_paths_under_asan_ubsan(tmp_path, harness, gateway, args):
    """Both mexFunction()s instrumented (the library itself is not): on a box without a GPU every call must end in a MEX error
    raised by the gateway — argument marshalling, context creation failure, clean-up — with no sanitizer report."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the error path under test is the no-GPU one (GPU runs of sanitizer builds are not made)")
    libdir = os.path.join(ROOT, "amaranth_twstft_amd")
    exe = tmp_path / harness
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", *SAN, "-I" + os.path.join(ROOT, "tests", "cpu", "mex_fake"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpu", harness + ".cpp"), os.path.join(ROOT, "mex", gateway + ".cpp"), "-L" + libdir, "-ltwstft_hip",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    np.zeros(40000 * 4, dtype=np.int16).tofile(tmp_path / "cap.bin")
    np.zeros(10000, dtype=np.uint8).tofile(tmp_path / "chips.bin")
    fill = dict(cap=str(tmp_path / "cap.bin"), chips=str(tmp_path / "chips.bin"), out=str(tmp_path / "out.bin"))
    r = subprocess.run([str(exe)] + [a.format(**fill) for a in args], capture_output=True, text=True, env=dict(ENV, ASAN_OPTIONS="detect_leaks=0"), timeout=600)
    _no_report(r)
    assert r.returncode == 3 and "twstft:create" in r.stderr and "no CPU fallback" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("real", [False, True])
def test_receiver_program_main_under_asan_ubsan(tmp_path, real):
    """apps/rxcomplex_hip.cpp (main() of the receiver programs) instrumented, on its argument / file error paths and — without a
    GPU — the library's refusal: exit code 1 and the program's message each time, no sanitizer report."""
    import torch
    libdir = os.path.join(ROOT, "amaranth_twstft_amd")
    exe = tmp_path / "rx_main"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", *SAN, *(["-DTWX_RX_REAL"] if real else []), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "apps", "rxcomplex_hip.cpp"), "-L" + libdir, "-ltwstft_hip", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0")
    run = lambda *a: subprocess.run([str(exe), *a], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=600)
    (tmp_path / "sdr.param").write_text("# c\nA N 100 0001186 2500 1250 2000 256 -18\nB S 101 0001186 2500 1250 2000 256 -18\n" + "A N 100 1 2500 1250 2000 256 -18\n" * 130)
    (tmp_path / "data.bin").write_bytes(b"\0" * 4096)
    for args, want in ((("a", "b", "c"), "usage:"), (("x.bin", "nope"), "no such parameter file"), (("x.bin",), "Data filename error")):
        r = run(*args)
        _no_report(r)
        assert r.returncode == 1 and want in r.stdout, r.stdout + r.stderr
    if not torch.cuda.is_available():
        r = run()                                               # 120 rows kept of 132 (nch_max), then twx_rx_create refuses: no device
        _no_report(r)
        assert r.returncode == 1 and "no HIP device" in r.stdout, r.stdout + r.stderr


def test_goranging_program_main_under_asan_ubsan(tmp_path):
    """apps/goranging_hip.cpp (main() of GoRanging, processing/CPP/main.cpp:773-807) instrumented, on its argument / file error paths
    and — without a GPU — the library's refusal: the program's messages, exit code 1, no sanitizer report."""
    import torch
    libdir = os.path.join(ROOT, "amaranth_twstft_amd")
    exe = tmp_path / "goranging_main"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", *SAN, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "apps", "goranging_hip.cpp"), "-L" + libdir, "-ltwstft_hip", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0")
    run = lambda *a: subprocess.run([str(exe), *a], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=600)
    np.zeros(40000 * 4, dtype=np.int16).tofile(tmp_path / "1670074501.bin")
    np.zeros(10000, dtype=np.uint8).tofile(tmp_path / "code.bin")
    (tmp_path / "empty.bin").write_bytes(b"")
    for args, want in (((), "data.bin code.bin [remote=0] [foffset=0.]"), (("1670074501.bin", "nope.bin"), "fcode read: FAIL"),
                       (("1670074501.bin", "empty.bin"), "fcode read: FAIL")):
        r = run(*args)
        _no_report(r)
        assert r.returncode == 1 and want in r.stdout, r.stdout + r.stderr
    if not torch.cuda.is_available():
        r = run("1670074501.bin", "code.bin", "1", "250.5")
        _no_report(r)
        assert r.returncode == 1 and "init error" in r.stdout and "20000 60000" in r.stdout, r.stdout + r.stderr
