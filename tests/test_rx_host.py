"""CPU: the parameter-file parser of the DLL/PLL receiver (twx_rx_parse_param, experiments/231001_DLL_PLL/rxcomplex.cpp:263-296)
against the oracle's restatement, and the receiver failing loudly without a GPU."""
import ctypes as C

import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import receiver
from oracle import twstft_oracle as orc

PARAM = """# chA_or_B Sic_or_Normal PRN_no. center_freq(Hz) chip_rate(kcps) LPF_cutoff(kHz) freq_search_range(Hz) freq_search_step(Hz) least_required_SNR(dB)
A N 100 0000186 2500 1250 64096 256 -18
B N 101 0000056 2500 1250 64096 256 -18
# commented A N 100 1 2500 1250 64096 256 -18
A S 105 -150000 2500 1250 1000;10;-30
B N 131 199999.5 2500 1250 199999 1 -99.9
C N 100 0000186 2500 1250 64096 256 -18
A X 100 0000186 2500 1250 64096 256 -18
A N 100 0000186 2500 1250 64096 256
A N 100 0000186 2500 1250 64096 256 -18 extra
A N 132 0000186 2500 1250 64096 256 -18
A N 100 0000186 1000 1250 64096 256 -18
A N 100 200000 2500 1250 64096 256 -18
A N 100 0000186 2500 1250 256 256 -18
A N 100 0000186 2500 1250 64096 256 -100
A N 7 -5.5 2500 1250 512 2 3
"""


def test_parameter_file_parser_matches_the_oracle(tmp_path):
    p = tmp_path / "sdr.param"
    p.write_text(PARAM)
    got = receiver.parse_param(str(p))
    want = orc.rx_parse_param(PARAM.splitlines(True))
    assert len(got) == len(want) == 4                      # rows 1, 2, the limits row, the short-code row (the ';' row passes the token count, not the scan)
    for g, w in zip(got, want):
        assert (g.ch.decode(), g.mode.decode(), g.pn, g.fc_init, g.kcps, g.fltkhz, g.frange, g.fstep, g.snr_min_db) == \
               (w["ch"], w["mode"], w["pn"], w["fc_init"], w["kcps"], w["fltkhz"], w["frange"], w["fstep"], w["snr_min_db"])
    lib = L.load()
    rows = (L.twx_rx_row * 2)()
    assert lib.twx_rx_parse_param(str(p).encode(), rows, 2) == 2            # capacity respected
    assert lib.twx_rx_parse_param(str(tmp_path / "missing").encode(), rows, 2) == -1
    assert b"no such parameter file" in lib.twx_rx_last_error(None)


def test_oracle_channel_setup_follows_the_program():
    import numpy as np
    row = orc.rx_parse_param(["A N 100 0000186 2500 1250 64096 256 -18\n"])[0]
    rng = np.random.default_rng(1)
    ci = orc.rx_channel_setup(row, rng.integers(0, 2, 100000), 10_000_000)
    assert (ci["clen"], ci["nlag"], ci["bps"], ci["nobs"], ci["nfft"]) == (100000, 28, 25, 400000, 1 << 20)
    assert (ci["range"], ci["step"]) == (65536.0, 256.0) and abs(ci["snr_min"] - 10 ** -1.8) < 1e-15
    assert ci["dat_name"] == "chA.pn100.2500kcps.dat"
    assert ci["log_set"] == "set param   : Ch. A, PRN#100,      186 2500  2500 65536   256   0\n"
    short = orc.rx_channel_setup(orc.rx_parse_param(["B N 7 -5.5 2500 1250 512 2 3\n"])[0], rng.integers(0, 2, 10000), 10_000_000)
    assert (short["clen"], short["nlag"], short["bps"], short["nobs"], short["nfft"]) == (10000, 14, 250, 40000, 1 << 17)


def test_receiver_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.TwxError, match="no HIP device|HIP"):
        receiver.Receiver([receiver.make_row("A", 100, 186.0, 2000.0, 256.0, -18.0, code=[0, 1] * 50000)])


@pytest.mark.parametrize("prog", ["rxcomplex_hip", "rx_hip"])
def test_receiver_programs_argument_and_error_paths(tmp_path, prog):
    """apps/rxcomplex_hip.cpp, the command-line drop-ins for ./rxcomplex and ./rx: the usage text (rxcomplex.cpp:175-180), the
    parameter-file and data-file errors (:213-217,254-255) with the programs' own words and exit code, and — here, without a GPU, or
    with one but without the code file — a loud failure from the library, never a silent run."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apps", "bin", prog)
    assert os.path.exists(exe), "build with make -C amaranth_twstft_amd/csrc (target apps)"
    run = lambda *a: subprocess.run([exe, *a], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    r = run("a", "b", "c")
    assert r.returncode == 1 and r.stdout.splitlines()[0] == "usage:" and "out_path param_file" in r.stdout
    r = run("nodata.bin", "noparam")
    assert r.returncode == 1 and r.stdout.splitlines() == ["nodata.bin", "no such parameter file : noparam"]
    (tmp_path / "sdr.param").write_text("A N 100 0001186 2500 1250 2000 256 -18\n")
    r = run("nodata.bin")
    assert r.returncode == 1 and r.stdout.splitlines() == ["nodata.bin", "Data filename error"]
    (tmp_path / "data.bin").write_bytes(b"\0" * 1000)
    r = run()                                                   # defaults ./data.bin and sdr.param (:185,208)
    assert r.returncode == 1 and r.stdout.splitlines()[0] == "./data.bin"
    assert "no HIP device" in r.stdout or "Code filename error" in r.stdout
    (tmp_path / "sdr.param").write_text("# nothing usable\nA N 100 0001186 1000 1250 2000 256 -18\n")
    r = run()
    assert r.returncode == 1 and "no usable row" in r.stdout


def test_parameter_file_parser_fuzz(tmp_path):
    """5000 generated parameter lines — well-formed rows with values around every limit of rxcomplex.cpp:288, and mutations of them
    (tokens dropped / added, separators of the token rule vs the scan rule, comment marks, lower case, numbers written as floats,
    with exponents, with junk behind them, empty and very long lines) — through twx_rx_parse_param and through the oracle's
    restatement: the same rows, in the same order, with the same values."""
    import numpy as np
    rng = np.random.default_rng(4711)
    lines = []
    def num(lo, hi, edge):
        v = float(rng.choice(edge)) if rng.integers(0, 3) == 0 else float(rng.uniform(lo, hi))
        style = int(rng.integers(0, 5))
        if style == 0: return "%d" % int(v)
        if style == 1: return "%.3f" % v
        if style == 2: return "%07d" % int(abs(v)) if v >= 0 else "-%06d" % int(abs(v))
        if style == 3: return "%g" % v
        return "%.2e" % v
    for _ in range(5000):
        wild = rng.integers(0, 4) == 0                                     # a quarter of the rows draw every field from the hostile sets
        toks = [str(rng.choice(["A", "B", "A", "B", "C", "a", ""] if wild else ["A", "B"])), str(rng.choice(["N", "S", "N", "X", "n"] if wild else ["N", "N", "S"])),
                num(-5, 140, [0, 99, 100, 131, 132, -1]) if wild else "%d" % int(rng.integers(0, 132)),
                num(-250000, 250000, [-200000, 199999, 200000, -200001, 0]) if wild else num(-199000, 199000, [-200000, 199999, 0]),
                str(rng.choice(["2500", "2500", "2500", "1000", "2500.0", "25e2"] if wild else ["2500"])), num(0, 3000, [1250]),
                num(-10, 250000, [0, 199999, 200000, 256, 1]) if wild else num(300, 190000, [199999, 65536]), num(0, 70000, [256, 1, 0]) if wild else num(1, 290, [256, 1]),
                num(-120, 30, [-100, -99.9, -18])]
        m = int(rng.integers(0, 12)) if rng.integers(0, 3) == 0 else 11
        if m == 0: toks = toks[:-1]
        elif m == 1: toks.append("extra")
        elif m == 2: toks[int(rng.integers(2, 9))] += "x"
        elif m == 3: toks.insert(int(rng.integers(0, 9)), "")
        sep = " " if m != 4 else str(rng.choice([";", "  ", " ;", "\t"]))
        line = sep.join(toks)
        if m == 5: line = "#" + line
        elif m == 6: line = " " + line
        elif m == 7: line = ""
        elif m == 8: line = line + " " * int(rng.integers(1, 150))
        lines.append(line + str(rng.choice(["\n", "\n", "\r\n"])))
    text = "".join(lines)
    p = tmp_path / "fuzz.param"
    p.write_bytes(text.encode())
    lib = L.load()
    rows = (L.twx_rx_row * 6000)()
    n = lib.twx_rx_parse_param(str(p).encode(), rows, 6000)
    want = orc.rx_parse_param(text.splitlines(True))
    assert n == len(want) and n > 300, (n, len(want))
    for i, w in enumerate(want):
        g = rows[i]
        assert (g.ch.decode(), g.mode.decode(), g.pn, g.fc_init, g.kcps, g.fltkhz, g.frange, g.fstep, g.snr_min_db) == \
               (w["ch"], w["mode"], w["pn"], w["fc_init"], w["kcps"], w["fltkhz"], w["frange"], w["fstep"], w["snr_min_db"]), (i, w)
