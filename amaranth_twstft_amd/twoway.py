"""Two-way combination of the four per-code delay series (host side, no GPU work).

Restates the arithmetic of acquisition/go_1s.m:83-268 on arrays instead of ``.mat`` files on disk:
for one session there are four result sets — station OP {local loop-back, remote} and station LTFB
{local, remote} — each the ``xval1 / indice1 / correction1 / SNR1r / SNR1i`` vectors written by the
tracked correlator (one element per 40-ms code, 25 codes/s).  The delivered product is

    res = 0.5*((opre-oplo)-(ltre-ltlo))      [ns]                       (go_1s.m:194)

plus the 1-second fitted values of the four series (``<MJD>.1s`` files, :251-268).

Names follow the script (oplo = OP local, opre = OP remote, ltlo/ltre = LTFB).  Octave is 1-based;
index sets returned here are 0-based.  ``pkg load nan`` is active in the script, so mean/median/std
ignore NaN (:16) — numpy's nan-functions are used accordingly.  UNPINNED (Octave only, no recorded
result files in the reference repository).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def valid_codes(xval: np.ndarray) -> tuple[np.ndarray, bool]:
    """Codes carrying signal: ``k=find(abs(xval1)>max(abs(xval1))/2)``, drop the first 10 (275 ms) and the last,
    cut at the first gap (go_1s.m:85-92).  Returns (0-based indices, truncated?)."""
    a = np.abs(np.asarray(xval))
    k = np.nonzero(a > a.max() / 2)[0]
    kk = np.nonzero(np.diff(k) > 1)[0]
    if kk.size:
        return k[10:kk[0] + 1], True          # k(11:kk(1))
    return k[10:-1], False                    # k(11:end-1)


def delays_ns(indice1, correction1, k, fs: float = 5e6, N: int = 1) -> np.ndarray:
    """``(indice1(k)+correction1(k)/(2*N+1))/fs*1e9`` (go_1s.m:93,95,120,122)."""
    indice1, correction1 = np.asarray(indice1, dtype=float), np.asarray(correction1, dtype=float)
    return (indice1[k] + correction1[k] / (2 * N + 1)) / fs * 1e9


def cut_at_sample_loss(lo_ns: np.ndarray, jump_ns: float = 2.0) -> tuple[np.ndarray, int | None]:
    """Loop-back delay jumps > 2 ns mean the SDR dropped samples: keep what precedes (go_1s.m:99-107)."""
    kk = np.nonzero(np.abs(np.diff(lo_ns)) > jump_ns)[0]
    if kk.size:
        first = int(kk[0]) + 1                # Octave kk(1), 1-based
        if first > 1:
            return lo_ns[:first - 1], first
        return lo_ns, first
    return lo_ns, None


def snr_db(SNR1r, SNR1i, k, fs: float = 5e6) -> float:
    """``median(10*log10(abs(SNR1r(k)+SNR1i(k))*fs))`` (go_1s.m:124,174)."""
    s = np.abs(np.asarray(SNR1r, dtype=float)[k] + np.asarray(SNR1i, dtype=float)[k]) * fs
    return float(np.median(10 * np.log10(s)))


def _fit_eval(x, y, deg):
    """``[~,s]=polyfit(x,y,deg); s.yf`` — fitted values at the abscissae."""
    c = np.polyfit(x, y, deg)
    return np.polyval(c, x)


@dataclass
class TwoWay:
    res: np.ndarray                 # per-code two-way result, ns (outliers NaN)
    res2: np.ndarray | None         # same with the remote series replaced by their quadratic fits
    oplo: np.ndarray
    opre: np.ndarray
    ltlo: np.ndarray
    ltre: np.ndarray
    resmean: float
    resstd: float
    resmean25: float
    resstd25: float
    opslope: np.ndarray             # polyfit(t, opre, 1) with t in seconds
    ltslope: np.ndarray
    one_second: np.ndarray = field(default_factory=lambda: np.zeros((0, 5)))   # [seconds since start, oplo, opre, ltlo, ltre]


def combine(oplo, opre, ltlo, ltre, N: int = 1, codes_per_s: int = 25, outlier_ns: float = 5.0,
            unwrap: bool = True) -> TwoWay:
    """go_1s.m:176-268 from the four delay series in ns (already restricted to their valid codes).

    ``unwrap`` applies the script's ±200/(2N+1) ns code-ambiguity shifts (:214-217) exactly as written —
    including the second test ``res>median(res)-10`` that adds the step to almost every element.  The script
    runs these lines unconditionally, so they are ON by default (a drop-in must deliver the script's numbers:
    on the reference's own 240527 records the mean moves by 200/3 ns, tests/test_ref_archives.py);
    ``unwrap=False`` gives the plain ``0.5*((opre-oplo)-(ltre-ltlo))``.
    """
    oplo, opre, ltlo, ltre = (np.asarray(v, dtype=float) for v in (oplo, opre, ltlo, ltre))
    m = min(len(oplo), len(ltlo))                                    # :176-182
    oplo, opre, ltlo, ltre = oplo[:m], opre[:m], ltlo[:m], ltre[:m]
    res2 = None
    if len(opre) > 2 and len(ltre) > 2:                              # :183-190
        x = np.arange(1, len(opre) + 1, dtype=float)
        res2 = 0.5 * ((_fit_eval(x, opre, 2) - oplo) - (_fit_eval(x, ltre, 2) - ltlo))
        res2[np.abs(res2 - np.nanmedian(res2)) > outlier_ns] = np.nan
    res = 0.5 * ((opre - oplo) - (ltre - ltlo))                      # :192
    res[np.abs(res - np.nanmedian(res)) > outlier_ns] = np.nan       # :193-194
    if unwrap:                                                       # :214-217
        step = 200.0 / (2 * N + 1)
        res[res > np.nanmedian(res) + 10] -= step
        res[res > np.nanmedian(res) - 10] += step
    resmean, resstd = float(np.nanmean(res)), float(np.nanstd(res, ddof=1))          # :242,247
    w = codes_per_s
    box = np.convolve(res, np.ones(w) / w)[w - 1:len(res) + w - 1 - w]              # conv(...)(25:end-25)
    resmean25 = float(np.nanmean(box)) if box.size else float("nan")
    resstd25 = float(np.nanstd(box, ddof=1)) if box.size > 1 else float("nan")
    t = np.arange(len(opre), dtype=float) / w
    opslope = np.polyfit(t, opre, 1) if len(opre) > 2 else np.full(2, np.nan)        # :277-278
    ltslope = np.polyfit(t, ltre, 1) if len(opre) > 2 else np.full(2, np.nan)
    rows = []
    cpt = 0
    for k in range(1, len(opre) - w + 1, w):                          # k=1:25:length(opre)-25 (:254)
        x = np.arange(k - 1, k + w - 1, dtype=float) / w
        sl = slice(k - 1, k + w - 1)
        vals = [_fit_eval(x, s[sl], 1)[w // 2] for s in (oplo, opre, ltlo, ltre)]      # yf(13)
        rows.append([float(cpt)] + [float(v) for v in vals])
        cpt += 1
    return TwoWay(res, res2, oplo, opre, ltlo, ltre, resmean, resstd, resmean25, resstd25, opslope, ltslope,
                  np.array(rows).reshape(-1, 5))


def session(op_local: dict, op_remote: dict, lt_local: dict, lt_remote: dict, fs: float = 5e6, N: int = 1, min_codes: int = 102,
            **kw) -> TwoWay | None:
    """The per-session body of go_1s.m:83-182: select codes on the LOCAL records, apply the same index set to the
    REMOTE records of the same station, cut at sample losses/gaps, then :func:`combine`.  Each argument is a dict
    with ``xval1 indice1 correction1`` (the tracked correlator's output).  Returns None where the script skips the
    session: no more than ``min_codes`` usable loop-back codes (``if (length(oplo)>102)``, :102)."""
    k, _ = valid_codes(op_local["xval1"])
    oplo = delays_ns(op_local["indice1"], op_local["correction1"], k, fs, N)
    oplo, _ = cut_at_sample_loss(oplo)
    if not len(oplo) > min_codes:
        return None
    xr = np.abs(np.asarray(op_remote["xval1"])[k])
    kkk = np.nonzero(xr > xr.max() / 2)[0]
    gaps = np.nonzero(np.diff(kkk) > 1)[0]
    if gaps.size:                                                      # :113-120
        k = k[:gaps[0] + 1]
        if gaps[0] + 1 < len(oplo):
            oplo = oplo[:gaps[0] + 1]
    opre = delays_ns(op_remote["indice1"], op_remote["correction1"], k, fs, N)[:len(oplo)]
    k, _ = valid_codes(lt_local["xval1"])
    ltlo = delays_ns(lt_local["indice1"], lt_local["correction1"], k, fs, N)
    xr = np.abs(np.asarray(lt_remote["xval1"])[k])
    kkk = np.nonzero(xr > xr.max() / 2)[0]
    gaps = np.nonzero(np.diff(kkk) > 1)[0]
    if gaps.size:                                                      # :160-165
        k = k[:gaps[0] + 1]
        ltlo = ltlo[:gaps[0] + 1]
    if len(kkk) < len(ltlo):                                           # :166-170
        k = k[:kkk[-1] + 1]
        ltlo = ltlo[:kkk[-1] + 1]
    ltre = delays_ns(lt_remote["indice1"], lt_remote["correction1"], k, fs, N)
    return combine(oplo, opre, ltlo, ltre, N=N, **kw)


# --------------------------------------------------------------------------------------------
# files: the four result records of a session and the delivered <MJD>.1s text (go_1s.m:83-95,126-139,251-268)
# --------------------------------------------------------------------------------------------

def julian_day(year: float, month: float, day: float) -> float:
    """``julianDay`` of acquisition/go_1s.m:18-32 (the branch for dates after 1582; ``day`` may carry a fraction)."""
    branch = year + (month - 1.0) / 12.0 + day / 365.25
    if np.floor(month) < 3:
        month += 12.0
        year -= 1.0
    if branch >= 1582.78:
        return float(np.floor(year * 365.25) + np.floor(year / 400.0) - np.floor(year / 100.0) + np.floor(30.59 * (month - 2.0)) + day + 1721088.5)
    if branch >= 0.0:
        return float(np.floor(year * 365.25) + np.floor(30.59 * (month - 2.0)) + day + 1721086.5)
    return float(np.sign(year) * np.floor(abs(year) * 365.25) + np.floor(30.59 * (month - 2.0)) + day + 1721085.5)


def mjd_of_unix(ts: float) -> float:
    """Session date as the script forms it from the 10-digit Unix time in the LTFB file name (go_1s.m:130-133):
    ``datevec`` of the time stamp, ``julianDay(y,m,d+(h+mi/60+s/3600)/24)-2400000.5+0.5-8.4e-2``."""
    import datetime
    t = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=float(ts))
    sec = t.second + t.microsecond * 1e-6
    return julian_day(t.year, t.month, t.day + (t.hour + t.minute / 60.0 + sec / 3600.0) / 24.0) - 2400000.5 + 0.5 - 8.4e-2


def octave_num2str(x: float) -> str:
    """Octave ``num2str`` of a real scalar (the ``<MJD>.1s`` file name, go_1s.m:252): integers as ``%d``, otherwise
    ``%.Ng`` with N = floor(log10(|x|)) + 5 significant digits, at least 5."""
    if x == np.floor(x):
        return "%d" % int(x)
    nd = max(int(np.floor(np.log10(abs(x)))) + 5, 5)
    return ("%.*g" % (min(nd, 16), x)).strip()


def write_1s(directory: str, mjd: float, one_second: np.ndarray) -> str:
    """The ``<MJD>.1s`` file of go_1s.m:251-268: header line, then one row per second
    ``MJD+cpt/86400  OPlocal  OPremote  LTFBlocal  LTFBremote`` (``%f`` each, tab separated).
    ``one_second`` = :attr:`TwoWay.one_second` (column 0 = cpt).  Returns the path written."""
    import os
    path = os.path.join(directory, octave_num2str(mjd) + ".1s")
    with open(path, "w") as fo:
        fo.write("# MJD\t\tOPlocal\tOPremote\tLTFBlocal\tLTBBremote\n")            # header exactly as the script writes it
        for row in np.asarray(one_second, dtype=float).reshape(-1, 5):
            fo.write("%f\t%f\t%f\t%f\t%f\n" % (mjd + row[0] / 86400.0, row[1], row[2], row[3], row[4]))
    return path


def load_record(path: str) -> dict:
    """One result file of the tracked correlator (``save -mat … corr* df indic* SNR* code puissan* xval* moved*``,
    claudio_aligned_code_ranging_separate.m:207), optionally gzip'ed as the archive keeps them (``*.mat.gz``):
    flat vectors ``xval1 indice1 correction1 SNR1r SNR1i``."""
    import gzip
    import io
    from scipy.io import loadmat
    raw = gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()
    m = loadmat(io.BytesIO(raw))
    return {k: np.asarray(v).reshape(-1) for k, v in m.items() if not k.startswith("__")}


def session_files(root: str, op_local_name: str):
    """The four files of one session as go_1s.m finds them (:83,110-111,126-139,149-151): given the OP local record
    ``OP/<name>`` → OP remote = same name with ``local``→``remote`` and ``_2``→``_1``; LTFB local = first file of
    ``LTFB/`` starting with the first 21 characters of the name (prefix + 9 of the 10 time-stamp digits: the two sites
    start within seconds of each other); LTFB remote = that name with ``local``→``remote``, ``_1.``→``_2.`` (first 22
    characters).  Returns (paths dict, LTFB time stamp) or (None, None) when a file is missing."""
    import glob
    import os
    op_lo = os.path.join(root, "OP", op_local_name)
    op_re = os.path.join(root, "OP", op_local_name.replace("local", "remote").replace("_2", "_1"))
    lt = sorted(glob.glob(os.path.join(root, "LTFB", op_local_name[:21] + "*")))
    if not (os.path.exists(op_lo) and os.path.exists(op_re) and lt):
        return None, None
    nom = os.path.basename(lt[0])
    nomre = nom.replace("local", "remote").replace("_1.", "_2.")
    ltre = sorted(glob.glob(os.path.join(root, "LTFB", nomre[:22] + "*")))
    if not ltre:
        return None, None
    return dict(op_local=op_lo, op_remote=op_re, lt_local=lt[0], lt_remote=ltre[0]), float(nom[12:22])


def process_sessions(root: str, out_dir: str | None = None, fs: float = 5e6, N: int = 1, pattern: str = "lo*gz"):
    """The loop of go_1s.m:77-268 over ``root/OP/lo*gz``: per session load the four records, combine
    (:func:`session`), write ``<MJD>.1s``.  Returns [(mjd, TwoWay, path of the .1s file)]."""
    import glob
    import os
    out = []
    for p in sorted(glob.glob(os.path.join(root, "OP", pattern))):
        files, ts = session_files(root, os.path.basename(p))
        if files is None:
            continue
        recs = {k: load_record(v) for k, v in files.items()}
        if any("xval1" not in r for r in recs.values()):
            continue
        tw = session(recs["op_local"], recs["op_remote"], recs["lt_local"], recs["lt_remote"], fs=fs, N=N)
        if tw is None:
            continue
        mjd = mjd_of_unix(ts)
        out.append((mjd, tw, write_1s(out_dir or root, mjd, tw.one_second)))
    return out
