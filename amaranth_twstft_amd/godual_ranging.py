"""Script-level drop-in for processing/Octave/godual_ranging.m (file-in / delay-out contract).

    python -m amaranth_twstft_amd.godual_ranging [--datalocation DIR] [--codelocation DIR] [--remote 0|1] [--OP 0|1]

Same flow as the reference script (godual_ranging.m:57-133): every capture ``1*.bin`` in
``datalocation`` (int16 ``[I1 Q1 I2 Q2]``) is correlated window by window against the code
``n*.bin`` picked by the parity of OP+remote (:60); one TSV row per window goes to stdout (:74,96,98)
and ``<capture>.mat`` (``remote<capture>.mat`` when remote=1) holds the result vectors (:126-131).
Already processed captures are skipped (the idempotence rule of
acquisition/claudio_aligned_code_ranging_separate.m:119).  Environment variables ``OP``,
``processing_dir``, ``codelocation`` are honoured like in the newer reference scripts (:12-25).
All sample arithmetic runs in libtwstft_hip.so.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

from . import prn, results_io
from .correlator import Correlator, band_godual


def run(datalocation="./", codelocation="./codes/", remote=0, OP=0, fs=5e6, Nint=1, out=sys.stdout, device=-1):
    caps = sorted(glob.glob(os.path.join(datalocation, "1*.bin")))
    codes = sorted(glob.glob(os.path.join(codelocation, "n*.bin")) + glob.glob(os.path.join(codelocation, "n*.bin.gz")))
    if not codes:
        raise FileNotFoundError(f"no code file n*.bin in {codelocation}")
    codefile = codes[(OP + remote) % 2 % len(codes)]              # LTFB=odd OP=even (godual_ranging.m:60)
    chips = prn.read_code_file(codefile)
    done = []
    with Correlator(chips, fs=fs, Nint=Nint, var_ddof=1, device=device) as cor:     # Octave var (N-1)
        band = band_godual(fs, cor.n, remote=remote, OP=OP)
        for cap in caps:
            base = os.path.basename(cap)
            nom = os.path.join(datalocation, ("remote" if remote == 1 else "") + base.replace(".bin", ".mat"))
            if os.path.exists(nom) or os.path.exists(nom + ".gz"):
                out.write(f"{nom} already done\n")
                continue
            out.write(base + "\n")
            if remote != 1:                    # both channels of every window from one pass over the file (:91,95)
                both = cor.process_file(cap, n_channels=2, channel=-1, band=band)
                r1, r2 = both[0], both[1]
            else:
                r1, r2 = cor.process_file(cap, n_channels=2, channel=0, band=band), None
            for row in results_io.tsv_rows(r1, r2, fs, Nint):
                out.write(row)
            results_io.save_mat(nom, r1, r2, code=prn.chips_to_code(chips), remote=remote)
            done.append(nom)
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--datalocation", default=os.environ.get("processing_dir", "./"))
    ap.add_argument("--codelocation", default=os.environ.get("codelocation", "./codes/"))
    ap.add_argument("--remote", type=int, default=0)
    ap.add_argument("--OP", type=int, default=int(os.environ.get("OP", "0") or 0))
    ap.add_argument("--fs", type=float, default=5e6)
    ap.add_argument("--Nint", type=int, default=1)
    a = ap.parse_args(argv)
    run(a.datalocation, a.codelocation, a.remote, a.OP, a.fs, a.Nint)


if __name__ == "__main__":
    main()
