"""CPU: pin the oracle against outputs of the reference's own code (tests/golden/*.json)."""
import hashlib
import math
import re

import numpy as np
import pytest

from oracle import twstft_oracle as orc
from amaranth_twstft_amd import prn
from tests.helpers import load_golden, capture_from_desc, chips_for


def test_lfsr_matches_reference_nextstate():
    g = load_golden("prn_codes.json")
    for e in g["lfsr"]:
        chips = orc.lfsr_chips(e["bitlen"], e["taps"], e["n"])
        assert hashlib.sha256(bytes(chips)).hexdigest() == e["chips_sha256"]
        assert list(chips[:64]) == e["chips_head"]
        # host mirror (fast block form) and its final state
        assert np.array_equal(prn.lfsr_chips(e["bitlen"], e["taps"], e["n"]), chips)
        a = 1
        for _ in range(e["n"]):
            a = prn.nextstate(a, e["taps"], e["bitlen"])
        assert a == e["final_state"]


def test_code_files_regenerate_bit_exact():
    g = load_golden("prn_codes.json")
    assert len(g["files"]) == 18
    for e in g["files"]:
        assert e["regenerated_matches"]
        chips = chips_for(e["bitlen"], e["taps"], e["len"])
        assert hashlib.sha256(bytes(chips)).hexdigest() == e["sha256"], e["name"]
        assert list(chips[:48]) == e["head"] and list(chips[-16:]) == e["tail"]


def _rows_221207(raw, chips, fs, Nint):
    """Oracle restatement of the printed row of 221207 godual_ranging.py:135."""
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    n = len(code)
    freq = orc.freq_axis(fs, n)
    k = orc.band_numpy(freq, 0.0, 8000.0)
    temps = np.arange(n) / fs
    raw = raw.reshape(-1, 4)
    rows = []
    for p in range(raw.shape[0] // n):
        blk = raw[p * n:(p + 1) * n]
        d1 = orc.deinterleave(blk, 2, 0); d1 = d1 - d1.mean()
        d2 = orc.deinterleave(blk, 2, 1); d2 = d2 - d2.mean()
        r1 = orc.processing(d1, k, freq, temps, fcode, code, Nint=Nint, fs=fs, snr_rot=-2)
        # channel 2 is NOT mixed in that file (fft2tmp=np.fft.fft(d2), :66) and rotates by 0 (:122)
        r2 = orc.processing(d2, k, freq, temps, fcode, code, Nint=Nint, fs=fs, snr_rot=0, df=0.0)
        rows.append(((r1["indice"] - r2["indice"] + r1["correction"] - r2["correction"]) / fs / 3,
                     r1["df"], 10 * np.log10(r1["puissance"]), 10 * np.log10(r1["SNRi"] + r1["SNRr"]),
                     10 * np.log10(r2["SNRi"] + r2["SNRr"])))
    return rows


@pytest.mark.parametrize("case", ["c5k", "c10k", "c25k", "c100k", "c250k", "c500k"])
def test_oracle_vs_221207_ranging_rows(case):
    g = load_golden("ref221207_ranging.json")
    c = next(x for x in g["cases"] if x["name"] == case)
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    rows = _rows_221207(raw, chips, c["fs"], c["Nint"])
    assert len(rows) == c["nwin"]
    for mine, line in zip(rows, c["rows"]):
        f = line.split("\t")
        ref = [float(v) for v in f[1:]]
        assert abs(mine[0] - ref[0]) <= 1.0e-12, (mine, ref)      # printed to 1e-12 s
        assert abs(mine[1] - ref[1]) <= 0.05 + 1e-9
        for a, b in zip(mine[2:], ref[2:]):
            assert abs(a - b) <= 0.05 + 1e-9


@pytest.mark.slow
def test_oracle_vs_220830_op_rows():
    """experiments/220830_OP/godual_ranging_OP.py run by tools/make_golden.py: zero-mean 0/1 replica (:17-24), x3
    interpolation written with concatenate/fftshift (:50-57), complex peak sample printed (:73).  The oracle's
    make_code_variant + processing() on the first 1-s window of the same capture."""
    g = load_golden("ref220830_op_ranging.json")
    c = g["cases"][0]
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    fs, n = g["fs"], 2 * len(chips)
    code = orc.make_code_variant(chips, None, 2, unipolar=True, zero_mean=True)
    fcode = np.conj(np.fft.fft(code))
    freq = orc.freq_axis(fs, n)
    k = orc.band_numpy(freq, 0.0, 8000.0)
    d = orc.deinterleave(raw[:n], 2, 0)
    d = d - d.mean()
    r = orc.processing(d, k, freq, np.arange(n) / fs, fcode, code, Nint=g["Nint"], fs=fs)
    ref = c["rows"][0]
    assert r["indice"] == ref["indice"]
    assert abs(r["correction"] - ref["correction"]) < 1e-9
    assert abs(r["xval"] - complex(*ref["xval"])) <= 1e-9 * abs(complex(*ref["xval"]))


@pytest.mark.slow
@pytest.mark.parametrize("case", ["n2M", "n2M_loopback", "n5M_taps57_remote"])
def test_oracle_vs_221219_processing(case):
    g = load_golden("ref221219_processing.json")
    c = next(x for x in g["cases"] if x["name"] == case)
    chips, raw = capture_from_desc(c["synth"], c["input_sha256"])
    fs = c["fs"]
    code = orc.make_code(chips, 2)
    fcode = orc.make_fcode(code)
    n = len(code)
    freq = orc.freq_axis(fs, n)
    k = orc.band_numpy(freq, 0.0, 8000.0)
    if case == "n5M_taps57_remote":                 # LTFB code (taps 57) received ~50 kHz off: the remote band of godual_ranging.m:88
        k = orc.band_godual(freq, remote=1, OP=0)
    assert [int(k[0]), int(k[-1])] == c.get("band_k", [int(k[0]), int(k[-1])])
    temps = np.arange(n) / fs
    d = orc.deinterleave(raw, 1, 0)
    d = d - d.mean()
    r = orc.processing(d, k, freq, temps, fcode, code, Nint=c["Nint"], fs=fs, fine_freq=True)
    ref = c["ref"]
    assert r["indice"] == ref["indice"]
    assert abs(r["correction"] - ref["correction"]) < 1e-9
    assert abs(r["df"] - ref["df"]) < 1e-9
    for key in ("SNRr", "SNRi", "puissance", "puissancecode", "puissancenoise"):
        assert abs(r[key] - ref[key]) <= 1e-9 * abs(ref[key]), key


def test_snr_identity_used_by_device_path():
    """mean(yincode) = (z[i-1]+z[i]+z[i+1])/M and mean|yint|^2 = mean|y|^2/R^2."""
    rng = np.random.default_rng(3)
    chips = chips_for(13, 27, 2500)
    code = orc.make_code(chips, 2)
    n = len(code)
    y = np.roll(code, 777) * 3 + rng.standard_normal(n) + 1j * rng.standard_normal(n)
    fcode = orc.make_fcode(code)
    ffty = np.fft.fft(y)
    z = orc.xcorr_interp(ffty, fcode, 1)
    ind, *_ = orc.peak_refine(z)
    SNRr, SNRi, pcode, pnoise = orc.snr_wipeoff(ffty, code, ind, 1, rot=-1, ddof=0)
    M = 3 * n
    mean = (z[ind - 1] + z[ind] + z[(ind + 1) % M]) / M
    p2 = np.mean(np.abs(y) ** 2) / 9
    var = p2 - abs(mean) ** 2
    assert abs(mean.real ** 2 / var - SNRr) < 1e-10 * SNRr
    assert abs(mean.imag ** 2 / var - SNRi) < 1e-10 * max(SNRi, 1e-30) + 1e-18
    assert abs(var - pnoise) < 1e-10 * pnoise


def test_claudio_convention_is_mirrored_godual():
    """ifft(F·conj(Y)) = conj(reverse(ifft(conj(F)·Y))) — the cross-check SURVEY.md §8c names."""
    rng = np.random.default_rng(4)
    chips = chips_for(13, 27, 2500)
    code = orc.make_code(chips, 2)
    n = len(code)
    fs = 5e6
    temps = np.arange(n) / fs
    d = np.roll(code, 1500) * 5 * np.exp(2j * np.pi * 300.0 * temps) + rng.standard_normal(n) + 1j * rng.standard_normal(n)
    d = d - d.mean()
    freq = orc.freq_axis(fs, n)
    g = orc.processing(d, None, freq, temps, orc.make_fcode(code), code, Nint=1, fs=fs, df=300.0)
    c = orc.processing_claudio(d, 300.0, temps, orc.make_fcode(code, "claudio"), code, Nint=1)
    M = 3 * n
    assert c["indice"] == (M - g["indice"]) % M
    assert abs(c["xval"] - np.conj(g["xval"])) < 1e-9 * abs(g["xval"])
    assert abs(c["correction"] + g["correction"]) < 1e-9


def test_tracked_loop_restatement_aligns_the_window():
    """oracle.ranging_tracked (claudio_aligned_code_ranging_separate.m:143-205): carrier found by search_df, the first
    code re-aligns the window so the peak sits at sample 21 (:176), later codes stay aligned, the carry-over keeps
    the code count equal to the number of whole code periods."""
    from amaranth_twstft_amd import synth
    nchips, n, ncodes = 10000, 20000, 120
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=1500 * 256, fstep=synth.fstep_for_df(30.0, 5e6), phi0=5, amp=500,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=3)
    raw = synth.synth_channel(n * ncodes, chips, 2, p)
    out = orc.ranging_tracked(raw, chips, fs=5e6, ls_samples=50 * n)
    assert out["kbon"] > 0 and abs(out["df"][0] - 30.0) < 2.5
    assert out["moved"] == [1] and abs(out["movedval"][0] - ((n - 1500) + 1 + 1 / 3)) < 1.0
    ind = np.array(out["indice1"])
    assert ind[0] == 64.0 and np.all(np.abs(ind[1:] - 64.0 / 3) < 1e-9)          # raw 3N index after the move, /3 afterwards
    assert len(ind) == 2 * 50 - 1                                                  # 2 whole chunks; the move costs one code


def test_m_sequence_tap_search_known_answers(tmp_path):
    """tools/README.md:1-11 lists the 17-bit taps below 100 that mseq_calculator reports OK; common.py:32-57 m_seq_codes
    returns the same set in ascending order.  The matrix-order test must agree with stepping through the cycle."""
    assert prn.m_seq_codes(17, 8) == [9, 15, 33, 45, 51, 63, 65, 85]
    for n in (5, 6, 7, 8, 9):
        for code in range(1, 1 << n, 2):
            assert prn.lfsr_is_maximal(n, code) == (prn.lfsr_period(n, code) == (1 << n) - 1), (n, code)
    for n, taps in ((13, 27), (14, 43), (14, 57), (15, 3), (15, 17), (17, 9), (17, 15), (18, 39), (19, 63), (22, 3), (22, 57)):
        assert prn.lfsr_is_maximal(n, taps), (n, taps)           # every code family of the reference is maximal-length


def test_qpsk_code_file_layout(tmp_path):
    """common.py:59-73 with taps_b: a0 b0 a1 b1 ..., name prn<a>.<b>qpsk<bits>bits.bin; goqpsk.m:10-11 de-interleaves it."""
    import os
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        name = prn.write_prn_seq(14, 1000, 43, 57)
        assert name == "prn43.57qpsk14bits.bin"
        code = prn.read_code_file(name)
        assert np.array_equal(code[0::2], prn.lfsr_chips(14, 43, 1000)) and np.array_equal(code[1::2], prn.lfsr_chips(14, 57, 1000))
        assert prn.write_prn_seq(14, 10, 43) == "prn43bpsk14bits.bin"
    finally:
        os.chdir(cwd)


def test_against_the_compiled_reference_period_checker():
    """oracle/_ref/mseq_calculator = the reference's own tools/mseq_calculator.c compiled from where it lies
    (oracle/Makefile).  Its period and OK verdict against prn.lfsr_period / lfsr_is_maximal.  (17, 10): an even tap mask,
    the checker reports the length-1 cycle it falls into.)"""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "mseq_calculator")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mseq_calculator not built (needs /root/reference)")
    for n, taps in ((13, 27), (14, 43), (14, 57), (15, 3), (17, 9), (17, 15), (17, 33), (17, 11), (17, 13), (12, 83), (12, 9)):
        out = subprocess.run([exe, str(n), str(taps)], capture_output=True, text=True).stdout
        m = re.search(r"->\s*(\d+)/(\d+)(\s+OK)?", out)
        assert m, out
        period, full, ok = int(m.group(1)), int(m.group(2)), bool(m.group(3))
        assert full == (1 << n) - 1
        assert ok == prn.lfsr_is_maximal(n, taps), (n, taps, out)
        if taps & 1:
            assert period == prn.lfsr_period(n, taps), (n, taps, out)
