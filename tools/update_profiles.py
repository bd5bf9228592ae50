#!/usr/bin/env python3
"""Copy the summaries of one tools/prof_round.sh run (gpurun_out/<dir>) into profiles/ (newest CSV per pass)."""
import glob, json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r01f"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
fetch = newest(f"{R}/pmc_fetch/*/*_counter_collection.csv")
write = newest(f"{R}/pmc_write/*/*_counter_collection.csv")
s1 = newest(f"{R}/stats_1slot/*/*_kernel_stats.csv")
s3 = newest(f"{R}/stats_3slot/*/*_kernel_stats.csv")
out = json.loads(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "pmc", fetch, write]))
per = {k: {"read_bytes": int(2 * v["FETCH_SIZE_KiB_max"] * 1024), "write_bytes": int(v["WRITE_SIZE_KiB_max"] * 1024)} for k, v in out.items()}
mid = per.get("k_row_mid(dif)") or per["k_row_mid"]
doc = {"note": "per launch of 8 windows x 5e6 samples; read = 2 x FETCH_SIZE (gfx950 half-count correction, calibrated on k_sums/k_col_inv "
               "whose compulsory reads are exactly 160 MB / 960 MB), write = WRITE_SIZE; separate --pmc passes (rocprofv3 --pmc FETCH_SIZE / "
               "--pmc WRITE_SIZE); k_row_mid is the DIF/DIT form k_rowd<MID> (the default)",
       "kernel": "k_row_mid", "bytes_per_launch": mid["read_bytes"] + mid["write_bytes"], "per_kernel": per}
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc_raw.json", "w"), indent=1)
shutil.copy(s1, f"profiles/{tag}_kernel_stats.csv")
shutil.copy(s3, f"profiles/{tag}_kernel_stats_default_3slots.csv")
for src, dst in ((f"profiles/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats.md"),
                 (f"profiles/{tag}_kernel_stats_default_3slots.csv", f"profiles/{tag}_kernel_stats_default_3slots.md")):
    open(dst, "w").write(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "stats", src]).decode())
shutil.copy(f"{R}/bench_default.json", f"profiles/{tag}_bench_line.json")
shutil.copy(f"{R}/bench_1slot.json", f"profiles/{tag}_bench_line_1slot.json")
d = json.load(open(f"profiles/{tag}_bench_line.json"))
print(d["value"], d["roofline"], d.get("other_workload"))
print(open(f"profiles/{tag}_kernel_stats.md").read()[:900])
