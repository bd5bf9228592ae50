#!/usr/bin/env python3
"""Side-by-side cache counters of k_col_inv3 / k_rowd<BAND> and the stand-alone probe of k_col_inv's read shape (tools/bw_probe.hip `colinv`),
from the per-counter rocprofv3 --pmc passes of tools/gpu_r06a.sh (one counter per pass, program directly behind `--`).
    python tools/colinv_counters.py gpurun_out/r06a > profiles/r06_colinv_counters.md"""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06a"
COUNTERS = ["TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCP_TCC_READ_REQ_LATENCY_sum",
            "TCP_PENDING_STALL_CYCLES_sum", "TCC_TAG_STALL_sum", "TA_BUSY_avr", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
ROWS = [("probe", "k_colinv<16, HIP_vector_type<float, 2u>, 1, 0, 0>", "probe colinv W=16 (plain loads)"),
        ("probe", "k_colinv<16, HIP_vector_type<float, 2u>, 1, 80000, 0>", "probe colinv W=16, 80 KB LDS (2 WG/CU)"),
        ("kern", "k_col_inv3", "k_col_inv3 (8 windows x 3 phases)"),
        ("band", "float, 1, 448", "k_rowd<BAND> (8 windows)")]
def med(kind, key, c):
    try:
        rows = list(csv.DictReader(open(os.path.join(d, f"{kind}_{c}.csv"))))
    except OSError:
        return None
    v = sorted(float(r["Counter_Value"]) for r in rows if r.get("Counter_Name") == c and key in r["Kernel_Name"])
    return v[len(v) // 2] if v else None
print("| counter (median per launch) | " + " | ".join(r[2] for r in ROWS) + " |")
print("|---|" + "---|" * len(ROWS))
tab = {}
for c in COUNTERS:
    vals = [med(k, key, c) for k, key, _ in ROWS]
    tab[c] = vals
    print(f"| {c} | " + " | ".join("—" if v is None else f"{v:.4g}" for v in vals) + " |")
print()
print("| derived | " + " | ".join(r[2] for r in ROWS) + " |")
print("|---|" + "---|" * len(ROWS))
def ratio(a, b):
    return ["—" if (x is None or y is None or y == 0) else f"{x / y:.4g}" for x, y in zip(tab[a], tab[b])]
print("| bytes read from L2's memory side (TCC_EA0_RDREQ x 128 B), GB | " + " | ".join("—" if v is None else f"{v * 128 / 1e9:.3f}" for v in tab["TCC_EA0_RDREQ_sum"]) + " |")
print("| mean L1->L2 read latency, cycles (LATENCY / READ_REQ) | " + " | ".join(ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum")) + " |")
print("| L1 pending-stall cycles per read request | " + " | ".join(ratio("TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum")) + " |")
print("| L2 hit fraction (HIT / REQ) | " + " | ".join(ratio("TCC_HIT_sum", "TCC_REQ_sum")) + " |")
print("| L2 tag-stall cycles per request | " + " | ".join(ratio("TCC_TAG_STALL_sum", "TCC_REQ_sum")) + " |")
