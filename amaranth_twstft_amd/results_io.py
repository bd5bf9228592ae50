"""Result containers in the reference's formats, so the later analysis scripts consume them unchanged.

* ``save_mat``: the variable set of ``save -mat … corr* df1 df2 indic* SNR* code puissan*``
  (processing/Octave/godual_ranging.m:126-131) plus ``xval*`` (acquisition/
  claudio_aligned_code_ranging_separate.m:207); 1×nwin row vectors, ``indice`` 1-based as Octave
  stores it, MAT v5 via scipy.io (schema seen in experiments/220616_Besancon/1655300700.mat.gz).
* ``tsv_rows``: the per-window line printed by godual_ranging.m:96,98.
"""
from __future__ import annotations

import numpy as np


def _vec(results, attr, dtype=np.float64):
    return np.array([[getattr(r, attr) for r in results]], dtype=dtype)


def mat_dict(res1, res2=None, code=None, remote: int = 0) -> dict:
    """Variables of godual_ranging.m:126-131 for channel 1 (measurement) and optionally 2 (reference)."""
    d = {}
    for suffix, res in (("1", res1), ("2", res2)):
        if res is None:
            continue
        d["indice" + suffix] = _vec(res, "indice").astype(np.float64) + 1.0        # Octave 1-based
        d["correction" + suffix] = _vec(res, "correction")
        d["df" + suffix] = _vec(res, "df")
        d[f"SNR{suffix}r"] = _vec(res, "SNRr")
        d[f"SNR{suffix}i"] = _vec(res, "SNRi")
        d["puissance" + suffix] = _vec(res, "puissance")
        d[f"puissance{suffix}code"] = _vec(res, "puissancecode")
        d[f"puissance{suffix}noise"] = _vec(res, "puissancenoise")
        d["xval" + suffix] = _vec(res, "xval", np.complex128)
        d[f"xval{suffix}m1"] = _vec(res, "xvalm1", np.complex128)
        d[f"xval{suffix}p1"] = _vec(res, "xvalp1", np.complex128)
    if code is not None:
        d["code"] = np.asarray(code, dtype=np.float64).reshape(1, -1)
    return d


def save_mat(path: str, res1, res2=None, code=None, remote: int = 0) -> None:
    import scipy.io
    scipy.io.savemat(path, mat_dict(res1, res2, code, remote), format="5", do_compression=False)


def tsv_rows(res1, res2, fs: float, Nint: int):
    """Lines of godual_ranging.m:96 (two channels) or :98 (remote, one channel); header of :74."""
    yield "n\tdt1\tdf1\tP1\tSNR1\tdt2\tdf2\tP2\tSNR2\r\n"
    r = 2 * Nint + 1
    for p, a in enumerate(res1, start=1):
        # Octave: (indice1(p)-1+correction1(p))/fs/(2*Nint+1) with 1-based indice == 0-based indice here
        row = "%d\t%.12f\t%.3f\t%.1f\t%.1f" % (p, (a.indice + a.correction) / fs / r, a.df, 10 * np.log10(a.puissance),
                                                 10 * np.log10(a.SNRi + a.SNRr))
        if res2 is not None:
            b = res2[p - 1]
            row += "\t%.12f\t%.3f\t%.1f\t%.1f" % ((b.indice - b.correction) / fs / r, b.df, 10 * np.log10(b.puissance),
                                                  10 * np.log10(b.SNRi + b.SNRr))
        yield row + "\r\n"


def polyfit_residuals(results, fs: float, Nint: int, sign: int = +1):
    """``[a,b]=polyfit([1:n],(solution)/(2*Nint+1)/fs,2); std(…-b.yf); mean(…-b.yf)`` of
    processing/Octave/godual_ranging.m:104-113 with ``solution=indice-1+correction`` (:104,106; both channels use
    ``+correction`` there).  Returns (std with Octave's N-1 normalisation, mean) of the quadratic-fit residual in
    seconds; (nan, nan) for fewer than 4 windows."""
    n = len(results)
    if n < 4:
        return float("nan"), float("nan")
    sol = np.array([r.indice + sign * r.correction for r in results], dtype=np.float64) / (2 * Nint + 1) / fs
    x = np.arange(1, n + 1, dtype=np.float64)
    res = sol - np.polyval(np.polyfit(x, sol, 2), x)
    return float(np.std(res, ddof=1)), float(np.mean(res))


def residual_report(res1, res2, fs: float, Nint: int):
    """The four ``ans = …`` lines Octave prints for godual_ranging.m:105-113 (loop-back channel first)."""
    for res in ((res2, res1) if res2 is not None else (res1,)):
        sd, mean = polyfit_residuals(res, fs, Nint)
        yield "ans = %.4e\n" % sd
        yield "ans = %.4e\n" % mean


def tracked_mat_dict(out: dict, code=None) -> dict:
    """Variables the tracked script leaves in its workspace for ``save -mat … corr* df indic* SNR* code puissan*
    xval* moved*`` (acquisition/claudio_aligned_code_ranging_separate.m:207): per-code row vectors ``xval1
    indice1 correction1 SNR1r SNR1i puissance1``, per-chunk ``df``, ``moved``/``movedval``.  ``out`` is the dict
    returned by ``tracked.TrackedRanging.run`` (``indice1`` already carries the script's own bookkeeping)."""
    row = lambda v, dt=np.float64: np.asarray(v, dtype=dt).reshape(1, -1)
    d = {"xval1": row(out["xval"], np.complex128), "indice1": row(out["indice1"]), "correction1": row(out["correction1"]),
         "SNR1r": row(out["SNR1r"]), "SNR1i": row(out["SNR1i"]), "puissance1": row(out["puissance1"]),
         "df": row(out["df"]), "moved": row(out["moved"]), "movedval": row(out["movedval"]),
         "puissancecode": row([out.get("puissancecode", np.nan)]), "puissancenoise": row([out.get("puissancenoise", np.nan)])}
    if code is not None:
        d["code"] = np.asarray(code, dtype=np.float64).reshape(1, -1)
    return d


def save_tracked_mat(path: str, out: dict, code=None) -> None:
    from scipy.io import savemat
    savemat(path, tracked_mat_dict(out, code), do_compression=False)


# --------------------------------------------------------------------------------------------
# the C++ twin's container (processing/CPP/main.cpp:521-656) — what gofinal_ltfb.m:35-45 reads when the name has a "C"
# --------------------------------------------------------------------------------------------

def cpp_mat_name(capture_path: str, remote: int = 0) -> str:
    """Output name of processing/CPP/main.cpp:786-798: directory kept, ``remote`` prefixed when remote=1, the
    capture's ``.bin`` replaced by ``C.mat`` (``<capture>C.mat``)."""
    import os
    d, base = os.path.split(capture_path)
    stem = base[:-4] if base.endswith(".bin") else base
    return os.path.join(d, ("remote" if remote == 1 else "") + stem + "C.mat")


def cpp_mat_dict(res1, res2=None) -> dict:
    """Variable set of ``GoRanging::save`` (main.cpp:541-647), n x 1 column vectors: ``correction1`` already holds
    ``indice+corr`` with the 0-based peak index (:310), ``SNR1`` = 10*log10(SNRr+SNRi) (:355), ``puissance1code`` in dB
    (:343), ``df1``, ``puissance1`` (sum |y|^2 of the window, :287-295), complex ``xval1 xval1m1 xval1p1``; the same
    with suffix 2 for the reference channel when it was processed."""
    d = {}
    col = lambda v, dt=np.float64: np.asarray(v, dtype=dt).reshape(-1, 1)
    for suffix, res in (("1", res1), ("2", res2)):
        if res is None:
            continue
        d["correction" + suffix] = col([r.indice + r.correction for r in res])
        d["SNR" + suffix] = col([10 * np.log10(r.SNRr + r.SNRi) for r in res])
        d["df" + suffix] = col([r.df for r in res])
        d["puissance" + suffix] = col([r.puissance for r in res])
        d[f"puissance{suffix}code"] = col([10 * np.log10(r.puissancecode) for r in res])
        d["xval" + suffix] = col([r.xval for r in res], np.complex128)
        d[f"xval{suffix}m1"] = col([r.xvalm1 for r in res], np.complex128)
        d[f"xval{suffix}p1"] = col([r.xvalp1 for r in res], np.complex128)
    return d


def save_cpp_mat(capture_path: str, res1, res2=None, remote: int = 0) -> str:
    """Write ``<capture>C.mat`` (MAT v5, uncompressed like MAT_COMPRESSION_NONE).  ``puissance`` of the results must come
    from a context with ``var_ddof=0``; the C++ program stores the un-normalised window power sum |y|^2 (:295), which
    is ``puissance*N`` plus N*|mean(y)|^2 — callers that need that exact number multiply by the window length."""
    import scipy.io
    path = cpp_mat_name(capture_path, remote)
    scipy.io.savemat(path, cpp_mat_dict(res1, res2), format="5", do_compression=False)
    return path


def read_gofinal_table(path_or_lines) -> dict:
    """A per-second table of gofinal_op.m / gofinal_ltfb.m (experiments/230111_twstft_2M5/gofinal_ltfb.m:86-89, the consumer of this
    package's ``.mat`` output): a ``%`` header line, then rows ``Y m d H M S<TAB>delay<TAB>df1<TAB>SNR1<TAB>delay2<TAB>df2<TAB>SNR2
    <TAB>delayrem<TAB>df1rem<TAB>SNR1rem`` (``%.12f  %.3f  %.1f`` three times); a row may stop after SNR2 (no remote solution): its
    remote columns are NaN.  ``path_or_lines``: a file (``.gz`` allowed) or an iterable of text lines.  Returns arrays keyed
    ``unix`` (seconds, the date taken as UTC), ``delay df1 SNR1 delay2 df2 SNR2 delayrem df1rem SNR1rem`` and ``date`` (n x 6 ints)."""
    import calendar
    import gzip
    if isinstance(path_or_lines, (str, bytes)):
        opener = gzip.open if str(path_or_lines).endswith(".gz") else open
        with opener(path_or_lines, "rt") as f:
            lines = f.read().splitlines()
    else:
        lines = list(path_or_lines)
    names = ("delay", "df1", "SNR1", "delay2", "df2", "SNR2", "delayrem", "df1rem", "SNR1rem")
    cols = {k: [] for k in names}
    dates = []
    for ln in lines:
        ln = ln.rstrip("\r\n")
        if not ln.strip() or ln.lstrip().startswith("%"):
            continue
        parts = ln.split("\t")
        d = [int(x) for x in parts[0].split()]
        if len(d) != 6:
            raise ValueError(f"not a gofinal row: {ln[:60]!r}")
        vals = [float(x) for x in parts[1:] if x.strip()]
        if len(vals) not in (6, 9):
            raise ValueError(f"{len(vals)} value columns in {ln[:60]!r} (6 or 9 expected)")
        vals += [np.nan] * (9 - len(vals))
        dates.append(d)
        for k, v in zip(names, vals):
            cols[k].append(v)
    out = {k: np.asarray(v, dtype=np.float64) for k, v in cols.items()}
    out["date"] = np.asarray(dates, dtype=np.int64).reshape(-1, 6)
    out["unix"] = np.asarray([calendar.timegm(tuple(d) + (0, 0, 0)) for d in dates], dtype=np.float64)
    return out
