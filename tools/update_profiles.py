#!/usr/bin/env python3
"""Copy the summaries of one tools/history/prof_round2.sh run (gpurun_out/<dir>) into profiles/ (newest CSV per pass).

    python tools/update_profiles.py gpurun_out/r02p r02
"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02p"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import short  # noqa: E402

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
fetch = newest(f"{R}/pmc_fetch/*/*_counter_collection.csv")
write = newest(f"{R}/pmc_write/*/*_counter_collection.csv")
s1 = newest(f"{R}/stats_1slot/*/*_kernel_stats.csv")
s3 = newest(f"{R}/stats_3slot/*/*_kernel_stats.csv")
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
out = json.loads(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "pmc", fetch, write]))
per = {k: {"read_bytes": int(2 * v["FETCH_SIZE_KiB_max"] * 1024), "write_bytes": int(v["WRITE_SIZE_KiB_max"] * 1024)} for k, v in out.items()}
mid = per.get("k_row_mid(dif)") or per["k_row_mid"]
doc = {"note": "per launch of 8 windows x 5e6 samples; read = 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md §HBM, calibrated on "
               "k_sums/k_col_inv whose compulsory reads are exactly 160 MB / 960 MB), write = WRITE_SIZE; separate --pmc passes (rocprofv3 --pmc "
               "FETCH_SIZE / --pmc WRITE_SIZE, tools/history/prof_round2.sh); k_row_mid is the DIF/DIT form k_rowd<MID> (the default)",
       "source_commit": head, "kernel": "k_row_mid", "bytes_per_launch": mid["read_bytes"] + mid["write_bytes"], "per_kernel": per}
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc_raw.json", "w"), indent=1)
shutil.copy(s1, f"profiles/{tag}_kernel_stats.csv")
shutil.copy(s3, f"profiles/{tag}_kernel_stats_default_3slots.csv")
pairs = [(f"profiles/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats.md"),
         (f"profiles/{tag}_kernel_stats_default_3slots.csv", f"profiles/{tag}_kernel_stats_default_3slots.md")]
for name in ("aux", "caf"):
    src = glob.glob(f"{R}/stats_{name}/*/*_kernel_stats.csv")
    if src:
        shutil.copy(max(src, key=os.path.getmtime), f"profiles/{tag}_{name}_kernel_stats.csv")
        pairs.append((f"profiles/{tag}_{name}_kernel_stats.csv", f"profiles/{tag}_{name}_kernel_stats.md"))
for src, dst in pairs:
    open(dst, "w").write(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "stats", src]).decode())
shutil.copy(f"{R}/bench_default.json", f"profiles/{tag}_bench_line.json")
shutil.copy(f"{R}/bench_1slot.json", f"profiles/{tag}_bench_line_1slot.json")
for f in ("aux_rates.jsonl", "bw_probe.txt", "valu_probe.txt"):
    if os.path.exists(f"{R}/{f}"):
        shutil.copy(f"{R}/{f}", f"profiles/{tag}_{f}")
# SQ counters of the dominant kernels (sum over dispatches of the 8-window launch)
sq = defaultdict(lambda: defaultdict(float))
for d in ("pmc_sq_a", "pmc_sq_b"):
    for path in glob.glob(f"{R}/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_row_mid") or k.startswith("k_col_inv") or k.startswith("k_col_fwd") or k.startswith("k_row_band"):
                sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = {}
for k, c in sq.items():
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    rows[k] = {n: v for n, v in c.items()}
    rows[k]["valu_issue_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4)
    rows[k]["wait_any_share"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 4)
    rows[k]["wait_inst_any_share"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        rows[k]["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "rocprofv3 --pmc SQ_* (two passes, tools/history/prof_round2.sh), summed over the dispatches of one 8-window batch; shares are of SQ_WAVE_CYCLES "
                   "(quad-cycles), bank conflicts of SQ_LDS_IDX_ACTIVE", "source_commit": head, "kernels": rows}, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
d = json.loads(open(f"profiles/{tag}_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"], d.get("other_workload"))
print(open(f"profiles/{tag}_kernel_stats.md").read()[:900])
# the text must follow the numbers: DESIGN.md's key-number table is compared with the files just written (5 % tolerance)
import check_design  # noqa: E402
sys.exit(check_design.check())
