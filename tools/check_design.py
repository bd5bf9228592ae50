#!/usr/bin/env python3
"""Do the numbers DESIGN.md quotes still agree with the committed profiles?

DESIGN.md (current numbers) and profiles/EXPERIMENTS.md (the log of rounds 1-4) each carry one table between the markers ``<!-- key-numbers -->`` and ``<!-- /key-numbers -->``; every row is

    | what | value unit | profiles/<file>:<selector> |

with a selector of one of these forms

    <json key path with dots>                    bench-line style JSON (last line of the file), e.g. ``roofline.avg_ms``
    jsonl[<field>=<substring>[,<field>=...]].<key>   first line of a .jsonl file whose fields contain the substrings (integer fields: equal)
    stats[<kernel-name substring>].<column>      row of a rocprofv3 ``*_kernel_stats.csv`` (columns as in the CSV, e.g. AverageNs)

The quoted value may carry a scale suffix the tool understands (``ms`` vs ``AverageNs`` → x 1e-6, ``Gsample/s`` vs Msamples/s →
x 1e-3, ``us`` vs ns → x 1e-3).  Exit status 1 when a quoted number differs from the profile by more than 5 % —
``tools/update_profiles.py`` ends with this check, so refreshing profiles/ without updating the text fails loudly.

The driver's own record: a row that cites the builder's driver-style line ``profiles/rNN_bench_line.json:<key>`` is ALSO held against
``BENCH_rNN.json`` (written by the round-end driver on its own box, ``parsed.<key>``) whenever that file exists: more than 5 % apart
fails as well — a number quoted from a run the driver's measurement contradicts is not a current number (boxes differ by 1-3 %).  The
record of the round being built does not exist yet while it is built; the check reports which rounds it could cross-check.

    python tools/check_design.py            # check
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 0.05


def _json_path(obj, path):
    for k in path.split("."):
        obj = obj[int(k)] if isinstance(obj, list) else obj[k]
    return float(obj)


def lookup(source: str) -> float:
    fname, sel = source.split(":", 1)
    path = os.path.join(ROOT, fname)
    m = re.match(r"jsonl\[([^\]]+)\]\.(.+)$", sel)
    if m:
        conds, key = m.groups()                              # field=substring[,field=substring ...]
        pairs = [c.split("=", 1) for c in conds.split(",")]
        for line in open(path):
            line = line.strip()
            if line.startswith("{"):
                j = json.loads(line)
                if all((str(j.get(f, "")) == sub) if str(j.get(f, "")).lstrip("-").isdigit() else (sub in str(j.get(f, ""))) for f, sub in pairs):
                    return _json_path(j, key)
        raise KeyError(f"{source}: no line with {conds}")
    m = re.match(r"stats\[([^\]]+)\]\.(\w+)$", sel)
    if m:
        sub, col = m.groups()
        for r in csv.DictReader(open(path)):
            if sub in r.get("Name", r.get("Kernel_Name", "")):
                return float(r[col])
        raise KeyError(f"{source}: no kernel ~ {sub}")
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return _json_path(json.loads(lines[-1]), sel)


def driver_value(source: str):
    """parsed.<key> of BENCH_rNN.json for a source ``profiles/rNN_bench_line.json:<key>``; None when there is no such record."""
    m = re.match(r"profiles/r(\d\d)_bench_line\.json:(.+)$", source)
    if not m or m.group(2).startswith("cpu_baseline"):         # (the host's CPUs, not the GPU: its spread between boxes is quoted in the row itself)
        return None
    path = os.path.join(ROOT, "BENCH_r%s.json" % m.group(1))
    if not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
        return _json_path(j.get("parsed", j), m.group(2))
    except (KeyError, ValueError, TypeError, IndexError):
        return None


CROSSCHECKED = set()
SCALES = {("ms", "ns"): 1e-6, ("us", "ns"): 1e-3, ("Gsample/s", "M"): 1e-3, ("TB/s", "GB"): 1e-3}


DOCS = ("DESIGN.md", os.path.join("profiles", "EXPERIMENTS.md"))      # current numbers; the log of rounds 1-4


def check_one(doc: str, verbose=True) -> int:
    txt = open(os.path.join(ROOT, doc)).read()
    m = re.search(r"<!-- key-numbers -->(.*?)<!-- /key-numbers -->", txt, re.S)
    if not m:
        print(f"{doc} has no key-numbers table")
        return 1
    bad = 0
    n = 0
    for row in m.group(1).splitlines():
        cells = [c.strip() for c in row.strip().strip("|").split("|")]
        if len(cells) != 3 or not cells[2].startswith("`profiles/"):
            continue
        what, quoted, source = cells[0], cells[1], cells[2].strip("`")
        qm = re.match(r"\**([0-9.]+)\**\s*(\S*)", quoted)
        if not qm:
            continue
        val, unit = float(qm.group(1)), qm.group(2)
        ref = lookup(source)
        scale = 1.0
        if unit == "ms" and source.endswith("Ns"):
            scale = 1e-6
        elif unit == "us" and source.endswith("Ns"):
            scale = 1e-3
        elif unit == "Gsample/s":
            scale = 1e-3
        elif unit == "TB/s":
            scale = 1e-3
        got = ref * scale
        n += 1
        ok = abs(val - got) <= TOL * abs(got)
        if verbose or not ok:
            print(f"{'ok ' if ok else 'BAD'} {what[:60]:60s} {doc} {val:g} {unit:10s} profile {got:.6g}  ({source})")
        bad += not ok
        drv = driver_value(source)
        if drv is not None:
            CROSSCHECKED.add(source.split("_")[0])
            dgot = drv * scale
            dok = abs(val - dgot) <= TOL * abs(dgot)
            if verbose or not dok:
                label, key = "  ... against the driver's own record", source.split(":", 1)[1]
                print(f"{'ok ' if dok else 'BAD'} {label:60s} {doc} {val:g} {unit:10s} driver  {dgot:.6g}  (BENCH_r{source[10:12]}.json:parsed.{key})")
            bad += not dok
    if n < 8:
        print(f"{doc}: only {n} checkable rows found")
        return 1
    return 1 if bad else 0


def check(verbose=True) -> int:
    rc = max(check_one(d, verbose) for d in DOCS)
    if verbose:
        print("cross-checked against the driver's BENCH records of rounds:", sorted(CROSSCHECKED) or "none (no BENCH_rNN.json for the rounds the tables cite)")
    return rc


if __name__ == "__main__":
    sys.exit(check())
