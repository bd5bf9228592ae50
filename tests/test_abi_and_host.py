"""CPU: the C-ABI library loads and exports every symbol the header declares; host-side logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import correlator as cor
from amaranth_twstft_amd import dist as D
from oracle import twstft_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "twstft_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(twx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/twstft_hip.h but not exported"
    assert set(names) == set(L.SYMBOLS), (set(names) ^ set(L.SYMBOLS))
    assert lib.twx_abi_version() == L.TWX_ABI_VERSION == 6          # 6: sharded recording over twx_multi, TWX_OPT_FIR_MFMA / SELFCHECK, resampled window, twx_extra (header)


def test_struct_layouts_match_header():
    assert C.sizeof(L.twx_result) == 240
    assert C.sizeof(L.twx_config) == 88
    assert C.sizeof(L.twx_band) == 16
    assert C.sizeof(L.twx_info) == 40
    assert L.twx_result.df.offset == 176 and L.twx_result.zwin.offset == 64
    assert C.sizeof(L.twx_tracked_config) == 104 and L.twx_tracked_config.band_lo_hz.offset == 48
    assert C.sizeof(L.twx_tracked_code) == 56 and C.sizeof(L.twx_tracked_summary) == 56


def test_tracked_defaults_mirror_the_three_scripts():
    """twx_tracked_defaults = the constants that tell claudio_aligned_code_{ranging,re,lo}_separate.m apart (band :134-141 /
    lo :105-113, carrier rule, floor :134, 30-s skip :128), compared with the oracle's table of the same constants."""
    lib = L.load()
    for mode, code in L_MODES.items():
        for OP in (0, 1):
            cfg = L.twx_tracked_config()
            assert lib.twx_tracked_defaults(code, OP, 5e6, C.byref(cfg)) == 0
            m = orc.tracked_mode(mode, OP)
            assert (cfg.band_lo_hz, cfg.band_hi_hz) == m["band"]
            assert cfg.carrier == (1 if m["carrier"] == "chunk_band" else 0) and cfg.indice_floor == int(m["indice_floor"])
            assert cfg.skip_samples == int(m["skip_seconds"] * 5e6) and cfg.chunk_samples == 10_000_000
            assert (cfg.sps, cfg.nint, cfg.df_threshold) == (2, 1, 20.0)
    assert lib.twx_tracked_defaults(7, 0, 5e6, C.byref(cfg)) == -1 and lib.twx_tracked_create(None, None) == -1


L_MODES = {"ranging": 0, "re": 1, "lo": 2}


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.TwxError) as e:
        cor.Correlator(lfsr=(14, 43, 10000))
    assert e.value.status == -3 and "no CPU fallback" in str(e.value)


def test_bad_arguments_are_errors_not_crashes():
    lib = L.load()
    h = C.c_void_p()
    assert lib.twx_create(None, C.byref(h)) == -1
    cfg = L.twx_config()
    cfg.fs, cfg.sps, cfg.nint, cfg.n_chips = 5e6, 2, 7, 1000
    assert lib.twx_create(C.byref(cfg), C.byref(h)) == -1
    assert b"nint" in lib.twx_last_error(None)
    assert lib.twx_strerror(-2) == b"unsupported window length"
    assert lib.twx_get_info(None, None) == -1


def test_round3_entry_points_reject_bad_arguments():
    """The tracked / acquisition / tracking entry points answer misuse with a status, never with a crash (no GPU involved)."""
    lib = L.load()
    h = C.c_void_p()
    cfg = L.twx_tracked_config()
    assert lib.twx_tracked_defaults(L.TWX_TRK_RE, 1, 5e6, C.byref(cfg)) == 0 and (cfg.band_lo_hz, cfg.band_hi_hz) == (-108000.0, -92000.0)
    cfg.chunk_samples = 0
    assert lib.twx_tracked_create(C.byref(cfg), C.byref(h)) == -1 and b"chunk_samples" in lib.twx_tracked_last_error(None)
    assert lib.twx_tracked_file(None, b"x", 0, -1, None) == -1 and lib.twx_tracked_fetch(None, None, None, None, None) == -1
    assert lib.twx_tracked_search_df(None, None, 0, None) == -1 and lib.twx_tracked_context(None) is None
    lib.twx_tracked_destroy(None)
    assert lib.twx_acquire_cdev(None, None, 0.0, 1.0, 1.0, 0, 0, None) == -1
    st, r = L.twx_track_state(), L.twx_track_result()
    z = np.zeros(24 * 57)
    assert lib.twx_track_update(z.ctypes.data, z.ctypes.data, 25, 28, C.byref(st), C.byref(r)) == -1      # fs = 0
    st.fs, st.duration = 5e6, 0.004
    assert lib.twx_track_update(None, z.ctypes.data, 25, 28, C.byref(st), C.byref(r)) == -1
    assert lib.twx_track_update(z.ctypes.data, z.ctypes.data, 25, 28, C.byref(st), C.byref(r)) == 0 and r.updated == 0   # all-zero powers: peak at the edge
    assert lib.twx_track_epoch_dev(None, None, 0, 1, 0, 10, 25, 28, None, 1.0, None, None) == -1


def test_band_helpers_match_reference_band_definitions():
    fs = 5e6
    for n in (20000, 200000):
        f = orc.freq_axis(fs, n)
        assert np.array_equal(cor.freq_axis(fs, n), f)
        k = orc.band_godual(f, 0, 0)
        assert cor.band_godual(fs, n) == (k[0], k[-1])
        k = orc.band_godual(f, 1, 0)
        assert cor.band_godual(fs, n, remote=1, OP=0) == (k[0], k[-1])
        k = orc.band_godual(f, 1, 1)
        assert cor.band_godual(fs, n, remote=1, OP=1) == (k[0], k[-1])
        k = orc.band_numpy(f, 0.0, 8000.0)
        assert cor.band_numpy(fs, n) == (k[0], k[-1])


def test_delay_formula():
    r = cor.WindowResult(3935295, -0.25, 1j, 1j, 1j, np.zeros(7, complex), 0.0, -1, 0, 0, 0, 0, 0)
    # (indice-1+correction)/fs/(2*Nint+1) with Octave's 1-based indice (godual_ranging.m:96)
    assert abs(r.delay(5e6, 1) - (3935295 - 0.25) / 5e6 / 3) < 1e-18


def test_shard_windows_partition():
    for n in (0, 1, 7, 600, 601):
        for world in (1, 2, 3, 8):
            spans = [D.shard_windows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == D.max_shard(n, world) or n == 0
    assert D.shard_windows(600, 3, 8) == (225, 300)


def test_header_is_plain_c_and_client_links(tmp_path):
    """include/twstft_hip.h compiles as C99 and a C client links against the shared library (no C++/torch types)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "amaranth_twstft_amd")
    exe = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpu", "abi_smoke.c"), "-L" + libdir, "-ltwstft_hip",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    assert exe.exists()


def test_mex_gateway_builds_against_the_functional_fake(tmp_path):
    """mex/twstft_processing_mex.cpp + tests/cpu/mex_harness.cpp compile and link against tests/cpu/mex_fake/mex.h and the
    library (no MATLAB/Octave in the image); without a GPU the gateway's create call fails with the library's message."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = build_mex_harness(root, tmp_path)
    import torch
    if torch.cuda.is_available():
        return
    np.zeros(40000 * 2, dtype=np.int16).tofile(tmp_path / "cap.bin")
    np.zeros(10000, dtype=np.uint8).tofile(tmp_path / "chips.bin")
    r = subprocess.run([str(exe), "complex", str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"),
                        "2", "1", "df", "0", "5e6", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "twstft:create" in r.stderr and "no CPU fallback" in r.stderr


def build_mex_harness(root, tmp_path, harness="mex_harness", gateway="twstft_processing_mex"):
    import subprocess
    libdir = os.path.join(root, "amaranth_twstft_amd")
    exe = tmp_path / harness
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(root, "tests", "cpu", "mex_fake"),
                    "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpu", harness + ".cpp"),
                    os.path.join(root, "mex", gateway + ".cpp"), "-L" + libdir, "-ltwstft_hip",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    return exe


def test_tracked_mex_gateway_builds_and_fails_loudly_without_gpu(tmp_path):
    """mex/twstft_tracked_mex.cpp (one call per capture file = the whole tracked flow) on the functional fake mex.h."""
    import subprocess
    exe = build_mex_harness(ROOT, tmp_path, "mex_tracked_harness", "twstft_tracked_mex")
    import torch
    if torch.cuda.is_available():
        return
    np.zeros(40000 * 2, dtype=np.int16).tofile(tmp_path / "cap.bin")
    np.zeros(10000, dtype=np.uint8).tofile(tmp_path / "chips.bin")
    r = subprocess.run([str(exe), str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"), "lo", "0", "5e6", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "twstft:create" in r.stderr and "no CPU fallback" in r.stderr
    r = subprocess.run([str(exe), str(tmp_path / "cap.bin"), str(tmp_path / "chips.bin"), str(tmp_path / "out.bin"), "sideways", "0", "5e6", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "mode must be" in r.stderr


def _octave_statements(path):
    """Statements of an Octave file with comments and all whitespace removed (what a copy check compares)."""
    out = []
    for line in open(path, encoding="utf-8", errors="replace"):
        line = re.sub(r"[%#].*$", "", line)
        for st in re.split(r"[;\n]", line):
            st = re.sub(r"\s+", "", st)
            if st:
                out.append(st)
    return out


def test_octave_drivers_share_only_contract_strings_with_the_reference():
    """The Octave drivers under mex/ keep the reference scripts' CONTRACT (file patterns, result names, saved variables) but
    none of their text: count the statements that are character-identical to a statement of the reference's tracked scripts."""
    ref_dir = "/root/reference/acquisition"
    if not os.path.isdir(ref_dir):
        pytest.skip("reference tree not present (GPU box)")
    ref = set()
    for f in ("claudio_aligned_code_ranging_separate.m", "claudio_aligned_code_re_separate.m", "claudio_aligned_code_lo_separate.m"):
        ref.update(_octave_statements(os.path.join(ref_dir, f)))
    trivial = {"end", "else", "do", "continue", "return"}
    mine = [s for s in _octave_statements(os.path.join(ROOT, "mex", "claudio_tracked_hip.m")) if s not in trivial]
    same = [s for s in mine if s in ref]
    assert len(mine) > 60 and len(same) <= 2, same
    assert not os.path.exists(os.path.join(ROOT, "mex", "claudio_aligned_code_ranging_separate_hip.m"))


def test_acquisition_gate_formula():
    """rxcomplex.cpp:570-573."""
    locked, p = cor.Correlator.acquisition_gate(pk=0.5, px=10.0, snr_min=0.1, psbb=2.0)
    assert p == 1.0 and locked == ((1.1 * 1.0) > (0.1 * 10.0))
    locked, _ = cor.Correlator.acquisition_gate(pk=0.1, px=10.0, snr_min=0.1, psbb=2.0)
    assert not locked


def test_lowpass_taps_match_an_independent_windowed_sinc():
    """frontend.lowpass_taps (GNU Radio ``firdes.low_pass(1, fs, fc, tw, WIN_HAMMING)`` of experiments/2403/zmq_rx.py:208-215,
    not installable here) against scipy.signal.firwin with the same length, cut-off and window."""
    from scipy.signal import firwin
    from amaranth_twstft_amd.frontend import lowpass_taps
    for fs, fc, tw in ((70e6, 2.1e6, 0.4e6), (5e6, 1.0e6, 0.2e6)):
        h = lowpass_taps(fs, fc, tw)
        assert h.size % 2 == 1 and h.size == int(53.0 * fs / (22.0 * tw)) | 1
        g = firwin(h.size, fc, window="hamming", fs=fs)
        assert np.abs(h - g).max() < 1e-7 and abs(h.sum() - 1.0) < 1e-6


def test_plan_generator_covers_the_reference_code_lengths():
    """amaranth_twstft_amd/plans.py: a split N1 x N2 with valid stage radices for every window length the reference's code files
    give at 1, 2 and 4 samples per chip (experiments/221207_twoway_codes/codes/*, 231001_DLL_PLL/{0,1}.bin), the 1-ms
    plumbing window, power-of-two acquisition transforms and lengths with a factor 7 (a native 70 Msps x 1 s window)."""
    from amaranth_twstft_amd import plans
    lengths = [c * s for c in (2500, 5000, 10000, 25000, 50000, 100000, 250000, 500000, 2500000) for s in (1, 2, 4)] + [1 << 16, 1 << 20, 30000, 486000, 14000, 70_000_000]
    for n in lengths:
        ch = plans.choose(n)
        assert ch is not None, n
        cp, rp = ch
        assert cp["L"] * rp["L"] == n and rp["L"] % cp["W"] == 0
        for pl in (cp, rp):
            prod = 1
            for r in pl["radices"]:
                assert 2 <= r <= 25
                prod *= r
            assert prod == pl["L"]
        assert (cp["L"] // min(cp["radices"])) * cp["W"] <= cp["nt"] <= 1024 and rp["L"] // min(rp["radices"]) <= rp["nt"] <= 1024
    assert plans.choose(22000) is None and plans.choose(5001) is None        # a factor 11 / an odd length have no plan


@pytest.mark.slow
def test_hot_kernels_compile_without_register_spills(tmp_path):
    """The three-workgroups-per-CU column kernels live under a hard register cap (__launch_bounds__(448, 6): 80 VGPRs);
    a spill there costs 2x in time and no GPU test would notice.  Cross-compile the N1 = 625 column plan and read the
    kernels' resource records."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "c625.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                    os.path.join(root, "amaranth_twstft_amd", "csrc", "twx_inst_col.hip"), "-DTWX_PLAN=Plan<625,25,25>", "-DTWX_W=16",
                    "-DTWX_NT=448", "-DTWX_NO_F64", "-o", str(out)], check=True, capture_output=True)
    txt = out.read_text()
    recs = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt)
    seen = 0
    for name, vgpr, spill in recs:
        if ("k_col_inv3" in name) or ("k_col_fwd3" in name and "InI16" in name):
            seen += 1
            assert int(spill) == 0 and int(vgpr) <= 80, (name, vgpr, spill)
    assert seen >= 3


@pytest.mark.slow
def test_row_walking_kernel_fits_two_workgroups_per_cu(tmp_path):
    """k_rowd<MID> in fp32 walks the rows with the next row loaded ahead: it must stay inside 128 VGPRs WITHOUT spills (a spill
    puts scratch traffic into the in-order memory counter its waits rely on) and inside half a CU's LDS (two resident workgroups)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "r8000.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                    os.path.join(root, "amaranth_twstft_amd", "csrc", "twx_inst_row.hip"), "-DTWX_PLAN=Plan<8000,20,20,20>", "-DTWX_NT=448",
                    "-DTWX_PADQ=20", "-DTWX_NO_F64", "-o", str(out)], check=True, capture_output=True)
    txt = out.read_text()
    seen = 0
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "k_rowdINS_4PlanILi8000" in name and "EEEfLi2ELi448ELb0E" in name:   # float, MODE = ROW_MID, CHK = false (the shipped default form)
            seen += 1
            vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
            spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
            lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            assert vgpr <= 128 and spill == 0 and scratch == 0 and lds <= 80 * 1024, (name, vgpr, spill, scratch, lds)
    assert seen == 1


def test_bluestein_length_search_is_fp64_aware_and_bounded():
    """cpp_twin._smooth_len: the context is opened in fp64, so only fp64-capable pairs count; lengths beyond every plan pair fail
    at once (a 10-minute capture used to spin for minutes in a pure-Python search before raising)."""
    import time
    from amaranth_twstft_amd import cpp_twin, plans
    m = cpp_twin._smooth_len(2 * 200_000 - 1)
    assert m >= 399_999 and plans.choose(m, f64=True) is not None and plans._smooth(m)
    assert cpp_twin._smooth_candidates(10, 40) == [10, 12, 14, 16, 18, 20, 24, 28, 30, 32, 36, 40]
    for lo in (2 * 48_000_000 - 1, 2 * 120_000_000 - 1):          # 240-s capture: fp32-only pairs; 10 minutes: no pair at all
        t = time.time()
        with pytest.raises(ValueError):
            cpp_twin._smooth_len(lo)
        assert time.time() - t < 2.0


def test_plan_file_names_carry_radices_and_precision(tmp_path, monkeypatch):
    """A plug-in's file name says what it holds (length, stage radices, tile width, fp32-only or fp32+fp64, source hash): an
    fp32-only object can no longer be mistaken for the fp64 build of the same length."""
    from amaranth_twstft_amd import plans
    cp, rp = plans.choose(5000)
    h = plans.source_hash()
    a = os.path.basename(plans._plan_file("col", cp, True))
    b = os.path.basename(plans._plan_file("col", cp, False))
    r = os.path.basename(plans._plan_file("row", rp, True))
    assert a == "col_%d_%s_w%d_f64_%s.so" % (cp["L"], "x".join(map(str, cp["radices"])), cp["W"], h)
    assert b == a.replace("_f64_", "_f32_") and r.startswith("row_%d_" % rp["L"]) and r.endswith("_f64_%s.so" % h)
    lib = L.load()
    assert lib.twx_load_plan(os.fsencode(str(tmp_path / "col_25_5x5_w16_f64_0000000000.so"))) != 0      # another source hash: refused
    assert b"kernel sources" in lib.twx_last_error(None)


def test_design_quotes_the_committed_profiles():
    """DESIGN.md's key-number table (chain rate, roofline fraction, per-kernel averages, CAF, acquisition, aux kernels) against
    the files under profiles/ it names: more than 5 % apart fails (tools/check_design.py, also the last step of
    tools/update_profiles.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_design", os.path.join(ROOT, "tools", "check_design.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(verbose=False) == 0


def test_every_ctypes_struct_has_the_layout_of_the_header(tmp_path):
    """Each ``twx_*`` structure of amaranth_twstft_amd/_lib.py against include/twstft_hip.h as gcc lays it out: the size and the
    offset and size of every field (a C program generated from the ctypes definitions prints them).  A field renamed, re-ordered
    or re-typed on one side only would otherwise show up as garbage in some result far from here."""
    import inspect
    import subprocess
    structs = [(n, c) for n, c in inspect.getmembers(L, inspect.isclass) if issubclass(c, C.Structure) and n.startswith("twx_") and hasattr(c, "_fields_")]
    assert len(structs) >= 15
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "twstft_hip.h"', 'int main(void) {']
    for n, c in structs:
        lines.append(f'  printf("{n} %zu\\n", sizeof({n}));')
        for f in c._fields_:
            lines.append(f'  printf("{n}.{f[0]} %zu %zu\\n", offsetof({n}, {f[0]}), sizeof((({n}*)0)->{f[0]}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    want = {}
    for ln in out:
        p = ln.split()
        if len(p) == 2:
            want[p[0]] = (int(p[1]),)
        elif len(p) == 3:
            want[p[0]] = (int(p[1]), int(p[2]))
    for n, c in structs:
        assert want[n] == (C.sizeof(c),), (n, want[n], C.sizeof(c))
        for f in c._fields_:
            d = getattr(c, f[0])
            assert want[f"{n}.{f[0]}"] == (d.offset, d.size), (n, f[0], want[f"{n}.{f[0]}"], (d.offset, d.size))


def test_every_binding_has_the_argument_list_of_the_header():
    """The ctypes prototypes of _lib.SYMBOLS against the declarations of include/twstft_hip.h: argument count, and for every argument
    and the return value the class of its type (pointer / 32-bit integer / 64-bit integer / double / float / void).  The C ABI does
    not check what a caller pushes: a binding one argument short, or an int where the header says int64_t, works until it does not."""
    txt = open(os.path.join(ROOT, "include", "twstft_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    decls = dict()
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(twx_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", txt):
        decls[m.group(2)] = (m.group(1).strip(), m.group(3).strip())
    assert set(decls) == set(L.SYMBOLS), set(decls) ^ set(L.SYMBOLS)

    def c_class(t):
        t = t.strip()
        if "*" in t:
            return "ptr"
        t = re.sub(r"\bconst\b", "", t).strip()
        t = re.sub(r"\s+[A-Za-z_][A-Za-z0-9_]*(\[\d*\])?$", "", t).strip() if " " in t else t          # drop the parameter name
        return {"void": "void", "int": "i32", "int32_t": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "long long": "i64",
                "size_t": "i64", "double": "f64", "float": "f32"}.get(t, "?" + t)

    def py_class(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int: "i32", C.c_int32: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_uint64: "i64", C.c_longlong: "i64", C.c_size_t: "i64",
                C.c_double: "f64", C.c_float: "f32"}.get(t, "?" + repr(t))

    bad = []
    for name, (ret, args) in sorted(decls.items()):
        res, argtypes = L.SYMBOLS[name]
        cargs = [] if args in ("", "void") else [a for a in args.split(",")]
        cl = [c_class(a) for a in cargs]
        if any("[" in a for a in cargs):                                   # array parameters decay to pointers
            cl = ["ptr" if "[" in a else c for a, c in zip(cargs, cl)]
        pl = [py_class(t) for t in argtypes]
        if cl != pl:
            bad.append((name, cl, pl))
        rc = "ptr" if "*" in ret else c_class(ret + " x")
        if rc != py_class(res):
            bad.append((name, "return", rc, py_class(res)))
    assert not bad, bad
