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
    if n < 8:
        print(f"{doc}: only {n} checkable rows found")
        return 1
    return 1 if bad else 0


def check(verbose=True) -> int:
    return max(check_one(d, verbose) for d in DOCS)


if __name__ == "__main__":
    sys.exit(check())
