#!/usr/bin/env python3
"""Generate tests/golden/*.json by RUNNING THE REFERENCE's own Python code in this container.

Run from the repo root in the build container (needs /root/reference; it does not exist on the
GPU box, which only ever sees the committed JSON):

    python tools/make_golden.py            # all fixtures
    python tools/make_golden.py --skip-5m  # skip the 5 M-sample case (≈1 min)

What is imported from the reference (never copied):
  * amaranth_twstft/common.py            nextstate()           → LFSR state/chip vectors
  * experiments/221219_twoway/processing/godual_ranging.py     processing() return values
  * experiments/221207_twoway_codes/processing/godual_ranging.py  ranging() printed rows
  * experiments/220830_OP/godual_ranging_OP.py                    ranging() printed rows (zero-mean 0/1 replica)
  * the code files under experiments/** (data)                 → SHA-256 + prefix

Inputs are synthetic captures from amaranth_twstft_amd.synth (integer-only generator), so the
fixtures carry only generator parameters, an input SHA-256 and the reference's outputs.
"""
from __future__ import annotations

import argparse
import contextlib
import gzip
import hashlib
import io
import json
import os
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

from amaranth_twstft_amd import prn, synth  # noqa: E402


def load_ref_module(relpath: str) -> dict:
    """exec a reference script minus its trailing module-level ``ranging(...)`` call."""
    src = open(os.path.join(REF, relpath)).read().rstrip("\n").split("\n")
    assert src[-1].startswith("ranging("), src[-1]
    ns: dict = {"__name__": "ref_" + re.sub(r"\W", "_", relpath)}
    exec(compile("\n".join(src[:-1]), os.path.join(REF, relpath), "exec"), ns)
    return ns


def cplx(z):
    return [float(np.real(z)), float(np.imag(z))]


def gen_prn():
    sys.path.insert(0, os.path.join(REF, "amaranth_twstft"))
    import common  # reference module (stdlib only)
    out = {"lfsr": [], "files": []}
    for bitlen, taps, n in [(13, 27, 5000), (14, 43, 10000), (17, 9, 4000), (22, 3, 3000), (22, 57, 3000),
                            (5, 5, 40), (19, 39, 2000)]:
        a = 1
        chips = []
        for _ in range(n):
            chips.append(a % 2)
            a = common.nextstate(a, taps, bitlen)
        out["lfsr"].append({"bitlen": bitlen, "taps": taps, "n": n, "final_state": a,
                            "chips_sha256": hashlib.sha256(bytes(chips)).hexdigest(),
                            "chips_head": chips[:64]})
    files = []
    cdir = os.path.join(REF, "experiments/221207_twoway_codes/codes")
    for f in sorted(os.listdir(cdir)):
        files.append(os.path.join(cdir, f))
    files += [os.path.join(REF, "experiments/231001_DLL_PLL/0.bin"),
              os.path.join(REF, "experiments/231001_DLL_PLL/1.bin"),
              os.path.join(REF, "experiments/220706_TWSTFT/OP_prn22bpskcode0.bin"),
              os.path.join(REF, "experiments/220706_TWSTFT/LTFB_prn57bpsk22bits.bin")]
    # (bitlen, taps) of each file; the two noiselen25000 files are really bitlen 15
    # (experiments/221207_twoway_codes/README.md:14), the others follow their names.
    special = {"0.bin": (17, 9), "1.bin": (17, 15), "OP_prn22bpskcode0.bin": (22, 3),
               "LTFB_prn57bpsk22bits.bin": (22, 57),
               "noiselen25000_bitlen17_taps03.bin.gz": (15, 3),
               "noiselen25000_bitlen17_taps17.bin.gz": (15, 17)}
    for path in files:
        raw = (gzip.open(path).read() if path.endswith(".gz") else open(path, "rb").read())
        name = os.path.basename(path)
        if name in special:
            bitlen, taps = special[name]
        else:
            m = re.match(r"noiselen(\d+)_bitlen(\d+)_taps(\d+)", name)
            bitlen, taps = int(m.group(2)), int(m.group(3))
        mine = prn.lfsr_chips(bitlen, taps, len(raw))
        ok = bytes(mine) == raw
        out["files"].append({"name": name, "relpath": os.path.relpath(path, REF), "len": len(raw),
                             "bitlen": bitlen, "taps": taps, "sha256": hashlib.sha256(raw).hexdigest(),
                             "head": list(raw[:48]), "tail": list(raw[-16:]),
                             "regenerated_matches": bool(ok)})
        print(f"  {name}: len {len(raw)} LFSR({bitlen},{taps}) match={ok}")
    return out


def synth_desc(n, bitlen, taps, nchips, sps, chans):
    return {"n": n, "bitlen": bitlen, "taps": taps, "nchips": nchips, "sps": sps,
            "channels": [vars(c).copy() for c in chans]}


def gen_221219(skip_5m: bool):
    ns = load_ref_module("experiments/221219_twoway/processing/godual_ranging.py")
    fs = ns["fs"]
    cases = []
    # (name, bitlen, taps, nchips, df, delay_q8, amp, sigma, seed, band): band "numpy" = the script's own 2*(foffset±frange),
    # "remote" = the 80..120 kHz band of processing/Octave/godual_ranging.m:88 (carrier offset 40..60 kHz)
    specs = [("n2M", 22, 3, 1000000, 1780.75, 733211 * 256 + 77, 200, 400.0, 11, "numpy"),
             ("n2M_loopback", 22, 57, 1000000, 0.0, 1200345 * 256, 3000, 100.0, 12, "numpy")]
    if not skip_5m:
        specs.append(("n5M_C2", 22, 3, 2500000, 1780.75, 1311765 * 256, 200, 400.0, 7, "numpy"))
        specs.append(("n5M_taps57_remote", 22, 57, 2500000, 50000.0 + 377.5, 2718281 * 256 + 128, 250, 450.0, 57, "remote"))
    for name, bitlen, taps, nchips, df, delay_q8, amp, sigma, seed, bandname in specs:
        chips = prn.lfsr_chips(bitlen, taps, nchips)
        n = 2 * nchips
        p = synth.SynthParams(delay_q8=delay_q8, fstep=synth.fstep_for_df(df, fs), phi0=0x12345678, amp=amp,
                              noise_gain=synth.noise_gain_for_sigma(sigma), seed=seed, stream=0)
        raw = synth.synth_channel(n, chips, 2, p)
        code = np.repeat(chips.astype(np.int64), 2) * 2 - 1          # as reference :74-77
        fcode = np.conj(np.fft.fft(code))
        freq = np.linspace(-fs / 2, fs / 2, num=len(code), dtype=float)
        if bandname == "numpy":
            k = np.nonzero((freq < 2 * (0 + 8000)) & (freq > 2 * (0 - 8000)))[0]
        else:
            k = np.nonzero((freq < 120000) & (freq > 80000))[0]
        temps = np.array(range(0, len(code))) / fs
        d = raw[:, 0].astype(complex)
        d.imag = raw[:, 1]
        d -= np.mean(d)
        r = ns["processing"](d, k, freq, temps, fcode, code)        # REFERENCE CALL
        indice, correction, SNRr, SNRi, dftmp, puissance, pcode, pnoise = r
        cases.append({"name": name, "synth": synth_desc(n, bitlen, taps, nchips, 2, [p]),
                      "input_sha256": hashlib.sha256(raw.tobytes()).hexdigest(),
                      "band": "numpy(foffset=0,frange=8000)" if bandname == "numpy" else "godual_ranging.m:88 remote band 80..120 kHz",
                      "band_k": [int(k[0]), int(k[-1])], "Nint": ns["Nint"], "fs": fs,
                      "ref": {"indice": int(indice), "correction": float(correction), "SNRr": float(SNRr),
                              "SNRi": float(SNRi), "df": float(dftmp), "puissance": float(puissance),
                              "puissancecode": float(pcode), "puissancenoise": float(pnoise)}})
        print(f"  221219 {name}: indice {indice} corr {correction:.6f} df {dftmp:.6f}")
    return {"source": "experiments/221219_twoway/processing/godual_ranging.py:processing", "cases": cases}


def gen_221207():
    ns = load_ref_module("experiments/221207_twoway_codes/processing/godual_ranging.py")
    fs = ns["fs"]
    cases = []
    specs = [("c5k", 13, 27, 5000, 4, 843.75, 11), ("c10k", 14, 43, 10000, 4, -1210.5, 12),
             ("c25k", 15, 3, 25000, 3, 2000.0, 13), ("c100k", 17, 9, 100000, 2, 1780.75, 14),
             ("c250k", 18, 39, 250000, 2, -915.25, 15), ("c500k", 19, 39, 500000, 2, 433.0, 16)]
    for name, bitlen, taps, nchips, nwin, df, seed in specs:
        chips = prn.lfsr_chips(bitlen, taps, nchips)
        n = 2 * nchips
        chans = [synth.SynthParams(delay_q8=(n // 3 + 1157) * 256 + 100, fstep=synth.fstep_for_df(df, fs), phi0=1 << 29,
                                   amp=300, noise_gain=synth.noise_gain_for_sigma(500.0), seed=seed, stream=0),
                 synth.SynthParams(delay_q8=(n // 3) * 256, fstep=0, phi0=0, amp=3000,
                                   noise_gain=synth.noise_gain_for_sigma(100.0), seed=seed, stream=1)]
        raw = synth.synth_capture(n * nwin, chips, 2, chans)
        with tempfile.TemporaryDirectory() as td:
            cap = os.path.join(td, "1670074501.bin")
            codef = os.path.join(td, "code.bin")
            raw.tofile(cap)
            chips.tofile(codef)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ns["ranging"](cap, codef)                           # REFERENCE CALL
        rows = [ln for ln in buf.getvalue().split("\n") if re.match(r"^\d+: \d{4} ", ln)]
        assert len(rows) == nwin, buf.getvalue()
        cases.append({"name": name, "synth": synth_desc(n * nwin, bitlen, taps, nchips, 2, chans),
                      "nwin": nwin, "input_sha256": hashlib.sha256(raw.tobytes()).hexdigest(),
                      "Nint": ns["Nint"], "fs": fs, "rows": rows})
        print(f"  221207 {name}: {rows[0]!r}")
    return {"source": "experiments/221207_twoway_codes/processing/godual_ranging.py:ranging (stdout)",
            "row_format": "p: Y m d H M S \\t (indice1-indice2+corr1-corr2)/fs/3 [1e-12] \\t df1 [0.1] \\t "
                          "10log10(var(y1)) \\t 10log10(SNR1i+SNR1r) \\t 10log10(SNR2i+SNR2r)",
            "cases": cases}


def gen_220830():
    """experiments/220830_OP/godual_ranging_OP.py: the one-second numpy script of the OP station — zero-mean 0/1 replica
    (``code=np.repeat(code,2); code=code-np.mean(code)``, :17-24), coarse carrier + mix on channel 1, ×3 zero-padded
    interpolation written with concatenate/fftshift (:50-57), printed ``p: indice+correction -- xval`` (complex peak
    sample) per 1-s window (:73).  Hard-wired to fs = 5e6 samples per window, two channels."""
    ns = load_ref_module("experiments/220830_OP/godual_ranging_OP.py")
    fs = ns["fs"]
    bitlen, taps, nchips, nwin = 22, 3, 2500000, 2
    chips = prn.lfsr_chips(bitlen, taps, nchips)
    n = 2 * nchips
    assert n == int(fs)
    chans = [synth.SynthParams(delay_q8=3141592 * 256 + 64, fstep=synth.fstep_for_df(-2345.5, fs), phi0=987654321, amp=260,
                               noise_gain=synth.noise_gain_for_sigma(420.0), seed=83, stream=0),
             synth.SynthParams(delay_q8=1000003 * 256, fstep=0, phi0=0, amp=2500,
                               noise_gain=synth.noise_gain_for_sigma(120.0), seed=83, stream=1)]
    raw = synth.synth_capture(n * nwin, chips, 2, chans)
    with tempfile.TemporaryDirectory() as td:
        cap = os.path.join(td, "17h05.bin")
        codef = os.path.join(td, "code.bin")
        raw.tofile(cap)
        chips.tofile(codef)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ns["ranging"](cap, codef)                               # REFERENCE CALL
    rows = []
    for ln in buf.getvalue().split("\n"):
        m = re.match(r"^(\d+): (\d+)\+(\S+) -- \((\S+?)([+-][^+-]+(?:e[+-]?\d+)?)j\)$", ln)
        if m:
            rows.append({"p": int(m.group(1)), "indice": int(m.group(2)), "correction": float(m.group(3)),
                         "xval": [float(m.group(4)), float(m.group(5))], "line": ln})
    assert len(rows) == nwin, buf.getvalue()[-2000:]
    for r in rows:
        print(f"  220830 {r['line']}")
    return {"source": "experiments/220830_OP/godual_ranging_OP.py:ranging (stdout, channel 1)",
            "row_format": "p: indice1+correction1 -- xval1   (indice 0-based on the x3 grid; xval = prnmap01[indice1], complex)",
            "replica": "zero-mean 0/1 code, 2 samples per chip", "band": "numpy(foffset=0,frange=8000)", "Nint": 1, "fs": fs,
            "cases": [{"name": "op_2win", "synth": synth_desc(n * nwin, bitlen, taps, nchips, 2, chans), "nwin": nwin,
                       "input_sha256": hashlib.sha256(raw.tobytes()).hexdigest(), "rows": rows}]}


def gen_mat_schema():
    """Variable names / shapes / dtypes of result archives the reference repository keeps (the data itself cannot be
    regenerated: the captures are not in the repository) — the schema the later analysis scripts load."""
    import gzip, io
    from scipy.io import loadmat
    out = {}
    for rel in ("experiments/220616_Besancon/1655300700.mat.gz", "experiments/230315_analysis_100k/local1674402311.mat.gz"):
        m = loadmat(io.BytesIO(gzip.open(os.path.join(REF, rel)).read()))
        out[rel] = {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in m.items() if not k.startswith("__")}
    return {"note": "schema only (names, shapes, dtypes) of result files in the reference repository", "files": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-5m", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    todo = a.only.split(",") if a.only else ["prn", "221207", "221219", "220830", "mat"]
    if "mat" in todo:
        print("result-file schemas")
        json.dump(gen_mat_schema(), open(os.path.join(GOLD, "mat_schema.json"), "w"), indent=1)
    if "prn" in todo:
        print("PRN fixtures")
        json.dump(gen_prn(), open(os.path.join(GOLD, "prn_codes.json"), "w"), indent=1)
    if "221207" in todo:
        print("221207 ranging() rows")
        json.dump(gen_221207(), open(os.path.join(GOLD, "ref221207_ranging.json"), "w"), indent=1)
    if "220830" in todo and not a.skip_5m:
        print("220830 OP ranging() rows")
        json.dump(gen_220830(), open(os.path.join(GOLD, "ref220830_op_ranging.json"), "w"), indent=1)
    if "221219" in todo:
        print("221219 processing() values")
        json.dump(gen_221219(a.skip_5m), open(os.path.join(GOLD, "ref221219_processing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
