#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.

    python tools/summarize_prof.py stats  <kernel_stats.csv>  > profiles/rNN_kernel_stats.md
    python tools/summarize_prof.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> [samples_per_launch]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(?:twx::)?(k_\w+)(?:<(.*)>)?\(", name)
    if not m:
        return name.split("(")[0][:60]
    k, targs = m.group(1), m.group(2) or ""
    mode = ""
    if k == "k_row":
        mm = re.search(r">, (float|double), (\d),", targs)
        if mm:
            mode = {"0": "_store", "1": "_band", "2": "_mid"}[mm.group(2)] + ("_f64" if mm.group(1) == "double" else "")
    elif k == "k_rowd":
        mm = re.search(r">, (float|double), (\d),", targs)
        if mm:
            return "k_row" + {"1": "_band", "2": "_mid"}.get(mm.group(2), "") + ("_f64" if mm.group(1) == "double" else "") + "(dif)"
    elif k == "k_col_fwd3":
        mm = re.search(r">, (float|double), \d+, (\d),", targs)
        if mm:
            return "k_col_fwd" + {"0": "_mix", "1": "_square", "2": "_plain"}[mm.group(2)] + "(split)"
    elif k == "k_rowd_bandsum":
        return "k_row_band(sum)"
    elif k == "k_col_inv3":
        return "k_col_inv(split)"
    elif k == "k_col_fwd":
        mm = re.search(r">, (float|double), \d+, (\d),", targs)
        if mm:
            mode = {"0": "_mix", "1": "_square", "2": "_plain"}[mm.group(2)] + ("_f64" if mm.group(1) == "double" else "")
    return k + mode


def stats(path):
    rows = list(csv.DictReader(open(path)))
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in rows:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")


def pmc(fetch_csv, write_csv, samples_per_launch=None):
    acc = defaultdict(lambda: defaultdict(list))
    for path in (fetch_csv, write_csv):
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in sorted(acc.items()):
        if not k.startswith("k_"):
            continue
        f = c.get("FETCH_SIZE", [0.0])
        w = c.get("WRITE_SIZE", [0.0])
        # counters are in KiB (cdna_hip_programming.md §7); per launch = max over launches (full batches)
        out[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_max": max(f), "WRITE_SIZE_KiB_max": max(w)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(*sys.argv[2:])
