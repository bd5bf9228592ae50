"""Plan plug-ins: FFT pass kernels for transform lengths the library was not built with.

The reference correlates against whatever code file it finds (``code=fread(f,inf,'int8')``,
processing/Octave/godual_ranging.m:62-66), so the window length N = n_chips*sps is arbitrary.  The library does a
length-N transform as N1 x N2 (column pass x row pass) with kernels instantiated at COMPILE time per
length; it ships the pairs the reference's own code files need.  For any other N = 2^a 3^b 5^c 7^d this module

* picks a split N = N1*N2 and stage radices that fit the kernels' budgets (one butterfly task per thread, <= 1024
  threads, the exchange buffers in LDS, radices <= 25 so that a butterfly stays in registers),
* compiles csrc/twx_inst_col.hip / csrc/twx_inst_row.hip for them with hipcc (seconds per file) into
  ``amaranth_twstft_amd/plans/*.so``,
* and hands them to the library (``twx_load_plan``); the library also picks up that directory by itself at the next
  ``twx_create`` (C / MEX hosts need no Python at run time).

    python -m amaranth_twstft_amd.plans 5000 25000        # build what these window lengths need

``Correlator`` calls :func:`ensure` on its own when the library reports TWX_E_SIZE.
"""
from __future__ import annotations

import itertools
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
PLAN_DIR = os.environ.get("TWX_PLAN_DIR") or os.path.join(_HERE, "plans")
HIPCC = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
MAX_RADIX = 25
MAX_THREADS = 1024
MAX_N1 = 10240                    # a column of complex double, one per workgroup, still fits the 160 KB of LDS
LDS_BUDGET = 80 * 1024            # bytes per workgroup that still leave room for a second one on the CU
LDS_MAX = 160 * 1024               # bytes of LDS a workgroup can have
ROW_TABLES = 12 * 1024            # stage / ramp tables of the row kernels beside the exchange buffer (complex float)


def _smooth(n: int) -> bool:
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def _radix_ok(r: int) -> bool:
    return 2 <= r <= MAX_RADIX and _smooth(r)


def stage_radices(L: int, max_stages: int = 4):
    """All ways to write L as a product of 1..max_stages radices (each 2^a 3^b 5^c 7^d <= 25), best first:
    fewer stages, then the largest minimum radix (fewest tasks per stage)."""
    cands = [r for r in range(2, MAX_RADIX + 1) if _radix_ok(r) and L % r == 0]
    out = []
    for s in range(1, max_stages + 1):
        for combo in itertools.combinations_with_replacement(cands, s):
            p = 1
            for r in combo:
                p *= r
            if p == L:
                out.append(tuple(sorted(combo)))
    return sorted(set(out), key=lambda c: (len(c), -min(c), max(c)))


def row_plan(L: int):
    """Stage order and thread count of a row plan of length L, or None.  Preferred: the DIF/DIT form
    [R0 *] R * R (RowD, csrc/twx_fft.h: one workgroup barrier per transform instead of one per stage)."""
    if L % 2 or L < 4:
        return None
    best = None
    for combo in stage_radices(L, 4):
        if len(combo) < 2:
            continue
        rowd = False
        order = combo
        if len(combo) == 2 and combo[0] == combo[1] and combo[0] <= 64:
            rowd = True
        elif len(combo) == 3:
            for r in set(combo):                   # two equal radices R, the third one is R0
                rest = list(combo)
                rest.remove(r)
                if r in rest:
                    rest.remove(r)
                    order = (rest[0], r, r)
                    rowd = True
                    break
        tasks = L // min(order)
        if tasks > MAX_THREADS:
            continue
        padq = order[0]
        lds = (L + L // padq) * 8                 # complex float exchange buffer; fp64 contexts need twice that
        if lds > LDS_MAX - ROW_TABLES:
            continue
        f64 = 2 * lds <= LDS_MAX - 2 * ROW_TABLES
        nt = max(64, -(-tasks // 64) * 64)
        key = (not f64, not rowd, len(order), -min(order))
        if best is None or key < best[0]:
            best = (key, dict(L=L, radices=order, nt=nt, padq=padq, rowd=rowd, f64=f64))
    return best[1] if best else None


def col_plan(L: int, row_len: int, f64: bool = True):
    """Column plan of length L for rows of length ``row_len``: widest tile W in (16, 8, 4, 2, 1) that divides the row
    and keeps tasks*W <= 1024 threads and the exchange in LDS; two-stage plans preferred (their fp32 kernels exchange
    one component at a time: half the LDS).  ``f64``: budget the exchange for complex double as well (False when the row
    plan of the pair exists in fp32 only: the pair is fp32-only anyway and the tile may be twice as wide)."""
    for W in (16, 8, 4, 2, 1):
        if row_len % W:
            continue
        for combo in stage_radices(L, 4):
            # stage order: largest radix first keeps the first (global -> register) stage wide
            order = tuple(sorted(combo, reverse=True))
            tasks = (L // min(order)) * W
            if tasks > MAX_THREADS:
                continue
            lds = L * W * (16 if f64 else 8) if len(order) > 1 else 0
            if lds > LDS_MAX:
                continue
            nt = max(64, -(-tasks // 64) * 64)
            return dict(L=L, radices=order, W=W, nt=nt, f64=f64)
    return None


def choose(n: int, f64: bool = False):
    """(column plan, row plan) for a window of n samples, or None.  ``f64``: only pairs whose complex-double kernels fit
    the LDS as well (fp32 contexts take the widest column tile first, which for very long windows is an fp32-only pair:
    N = 7e7 runs 16.7 Gsample/s as 7000 x 10000 at W = 2, 11.9 as the fp64-capable 8750 x 8000 at W = 1, tools/n70_rate.py)."""
    if n < 4 or n % 2 or not _smooth(n):
        return None
    best = None
    for n2 in range(16, min(n, 10000) + 1, 2):
        if n % n2:
            continue
        n1 = n // n2
        if n1 > MAX_N1:
            continue
        rp = row_plan(n2)
        if rp is None:
            continue
        if f64 and not rp["f64"]:
            continue
        cp = col_plan(n1, n2, rp["f64"]) if n1 > 1 else None
        if cp is None:
            continue
        # widest tile first (HBM pieces of 128 B), column workgroups of at least a wave, the DIF/DIT row form, few stages,
        # long rows (fewer, fatter workgroups)
        underfilled = (cp["L"] // min(cp["radices"])) * cp["W"] < 64
        key = (-cp["W"], not rp["f64"], underfilled, not rp["rowd"], len(rp["radices"]) + len(cp["radices"]), -n2)   # fp32-only rows: one workgroup per CU
        if best is None or key < best[0]:
            best = (key, cp, rp)
    return (best[1], best[2]) if best else None


PLAN_SOURCES = ("twx_fft.h", "twx_kernels.h", "twx_plans.h", "twx_inst_col.hip", "twx_inst_row.hip")


def source_hash() -> str:
    """Hash of the kernel sources a plug-in is compiled from = the tag the library looks for (csrc/Makefile SRCHASH)."""
    import hashlib
    h = hashlib.sha1()
    for f in PLAN_SOURCES:
        with open(os.path.join(CSRC, f), "rb") as fd:
            h.update(fd.read())
    return h.hexdigest()[:10]


def _clean_stale():
    """Plug-ins of other source versions are dead weight (the library ignores them)."""
    if not os.path.isdir(PLAN_DIR):
        return
    tail = "_%s.so" % source_hash()
    for f in os.listdir(PLAN_DIR):
        if (f.endswith(".so") and not f.endswith(tail)) or (f.endswith(".so.nof64") and not f[:-6].endswith(tail)):
            try:
                os.unlink(os.path.join(PLAN_DIR, f))
            except OSError:
                pass


def _plan_macro(L, radices):
    return "Plan<%d,%s>" % (L, ",".join(str(r) for r in radices))


def _compile(src, out, defs):
    """hipcc one plug-in; returns the path of the file written.  A plan whose complex-double instantiation does not fit the
    160 KB of LDS is rebuilt for fp32 only and written under the ``_f32_`` name (the name says what the object holds); a
    ``.nof64`` marker beside it records the failed attempt so that :func:`_build` does not repeat it on every call."""
    os.makedirs(PLAN_DIR, exist_ok=True)
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        raise RuntimeError(f"{HIPCC} not found: plan plug-ins are compiled with hipcc (set HIPCC=...)")
    tmp = out + ".tmp%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           os.path.join(CSRC, src), "-o", tmp, "-L" + _HERE, "-ltwstft_hip", "-Wl,-rpath,$ORIGIN/.."] + defs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode and "local memory" in r.stderr and "double" in r.stderr and "-DTWX_NO_F64" not in defs:
        # the complex-double instantiation of a long plan does not fit the 160 KB of LDS: fp32 contexts only
        # (the code spectrum is then computed in fp32 as well, Ctx::make_code_spectrum)
        got = _compile(src, out.replace("_f64_", "_f32_"), defs + ["-DTWX_NO_F64"])
        with open(got + ".nof64", "w") as fd:
            fd.write("the complex-double instantiation of this plan exceeds the LDS; fp32 only\n")
        return got
    if r.returncode:
        raise RuntimeError("plan build failed:\n" + " ".join(cmd) + "\n" + r.stderr[-3000:])
    os.replace(tmp, out)
    return out


def _plan_file(kind: str, plan: dict, f64: bool) -> str:
    """File name of a plug-in: everything that decides what the object holds is in it — length, stage radices, tile width
    (columns), precision set (f64 = fp32 + complex-double kernels, f32 = fp32 only) and the hash of the kernel sources.
    The library only looks at the trailing _<hash>.so."""
    rad = "x".join(str(r) for r in plan["radices"])
    w = "_w%d" % plan["W"] if kind == "col" else ""
    return os.path.join(PLAN_DIR, "%s_%d_%s%s_%s_%s.so" % (kind, plan["L"], rad, w, "f64" if f64 else "f32", source_hash()))


def _build(kind: str, src: str, plan: dict, defs) -> str:
    want64 = bool(plan.get("f64", True))
    out64, out32 = _plan_file(kind, plan, True), _plan_file(kind, plan, False)
    if os.path.exists(out64):
        return out64                                   # holds the fp32 kernels as well
    if os.path.exists(out32) and (not want64 or os.path.exists(out32 + ".nof64")):
        return out32                                   # asked for, or the fp64 attempt is known not to fit (marker of _compile)
    _clean_stale()
    return _compile(src, out64 if want64 else out32, defs + ([] if want64 else ["-DTWX_NO_F64"]))


def build_col(cp) -> str:
    return _build("col", "twx_inst_col.hip", cp, ["-DTWX_PLAN=" + _plan_macro(cp["L"], cp["radices"]), "-DTWX_W=%d" % cp["W"], "-DTWX_NT=%d" % cp["nt"]])


def build_row(rp) -> str:
    return _build("row", "twx_inst_row.hip", rp, ["-DTWX_PLAN=" + _plan_macro(rp["L"], rp["radices"]), "-DTWX_NT=%d" % rp["nt"], "-DTWX_PADQ=%d" % rp["padq"]])


def builtin_plans(lib=None) -> dict:
    """{(kind, precision): {(L, W)}} of what the library holds right now (kind 0 = columns, 1 = rows; W = 0 for rows)."""
    import ctypes as C
    from . import _lib as L
    lib = lib or L.load()
    have = {}
    for kind in (0, 1):
        for precision in (0, 1):
            cnt = lib.twx_plan_lengths(kind, precision, None, None, 0)
            ls, ws = (C.c_int32 * max(cnt, 1))(), (C.c_int32 * max(cnt, 1))()
            lib.twx_plan_lengths(kind, precision, ls, ws, cnt)
            have[(kind, precision)] = {(ls[i], ws[i]) for i in range(cnt)}
    return have


def ensure(n: int, precision: int = 0, lib=None, verbose: bool = False):
    """Make a plan pair for a window of ``n`` samples available to the library; returns the plug-in files it loaded
    ([] when the library already had one).  Raises ValueError for lengths outside 2^a 3^b 5^c 7^d or the kernels' budgets."""
    from . import _lib as L
    lib = lib or L.load()
    if lib.twx_plan_source_hash().decode() != source_hash():
        raise RuntimeError("libtwstft_hip.so was built from other kernel sources than csrc/ holds now: rebuild it (make -C csrc) first")
    if lib.twx_plan_available(int(n), int(precision)):
        return []
    ch = choose(int(n), precision == 1)
    if ch is None:
        raise ValueError(f"no N1 x N2 plan for a window of {n} samples (needs n even, = 2^a 3^b 5^c 7^d, N2 <= 10000, N1 <= %d)" % MAX_N1 + "")
    cp, rp = ch
    # reuse what the library already has (built-in or loaded) for either half
    hp = builtin_plans(lib)
    have = {0: hp[(0, int(precision))], 1: hp[(1, int(precision))]}
    files = []
    if (cp["L"], cp["W"]) not in have[0]:
        if verbose:
            print("building column plan", cp, file=sys.stderr)
        files.append(build_col(cp))
    if (rp["L"], 0) not in have[1]:
        if verbose:
            print("building row plan", rp, file=sys.stderr)
        files.append(build_row(rp))
    for f in files:
        if lib.twx_load_plan(os.fsencode(f)) != 0:
            raise RuntimeError("twx_load_plan failed: " + (lib.twx_last_error(None) or b"").decode())
    if not lib.twx_plan_available(int(n), int(precision)):
        if precision == 1 and lib.twx_plan_available(int(n), 0):
            raise ValueError(f"window of {n} samples: the complex-double kernels of this plan do not fit the LDS, use precision='f32'")
        raise RuntimeError(f"plan plug-ins {files} loaded but no pair for n = {n} is usable")
    return files


def main(argv=None):
    args = list(sys.argv[1:] if argv is None else argv)
    if not args:
        print(__doc__)
        return 2
    for a in args:
        n = int(a)
        ch = choose(n)
        print(n, "->", ch)
        print("   loaded:", ensure(n, verbose=True))
    return 0


if __name__ == "__main__":
    sys.exit(main())
