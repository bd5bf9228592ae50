#!/usr/bin/env python3
"""Copy the summaries of a tools/prof_round6.sh run (gpurun_out/r06p) into profiles/ and check DESIGN.md's numbers against them.
    python tools/copy_profiles_r06.py"""
import json, os, shutil, subprocess, sys
O, P, tag = "gpurun_out/r06p/", "profiles/", "r06"
sys.path.insert(0, "tools")
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
out = json.loads(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "pmc", O + "pmc_fetch_counter_collection.csv", O + "pmc_write_counter_collection.csv"]))
per = {k: {"read_bytes": int(2 * v["FETCH_SIZE_KiB_max"] * 1024), "write_bytes": int(v["WRITE_SIZE_KiB_max"] * 1024)} for k, v in out.items()}
mid = per.get("k_row_mid(dif)") or per["k_row_mid"]
doc = {"note": "per launch of 8 windows x 5e6 samples; read = 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md section HBM), write = WRITE_SIZE; "
               "separate --pmc passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, tools/prof_round6.sh); k_row_mid is the DIF/DIT form k_rowd<MID> (the default)",
       "source_commit": head, "kernel": "k_row_mid", "bytes_per_launch": mid["read_bytes"] + mid["write_bytes"], "per_kernel": per}
json.dump(doc, open(P + "pmc_traffic.json", "w"), indent=1)
json.dump(out, open(P + f"{tag}_pmc_raw.json", "w"), indent=1)
for src, dst in (("stats_1slot_kernel_stats.csv", f"{tag}_kernel_stats"), ("stats_3slot_kernel_stats.csv", f"{tag}_kernel_stats_default_3slots"),
                 ("stats_wide_kernel_stats.csv", f"{tag}_wideband_f64_kernel_stats")):
    shutil.copy(O + src, P + dst + ".csv")
    open(P + dst + ".md", "w").write(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "stats", P + dst + ".csv"]).decode())
for src, dst in (("bench_default.json", "bench_line.json"), ("bench_1slot.json", "bench_line_1slot.json"), ("bench_sustained.json", "bench_line_sustained_400steps.json"),
                 ("bench_single_process_8ctx.json", "bench_line_single_process_8ctx.json"), ("bench_selfcheck_on.json", "bench_line_selfcheck_on.json"), ("selfcheck_soak.jsonl", "selfcheck_soak.jsonl"),
                 ("bench_2ranks_rccl_asked.json", "bench_line_2ranks_rccl_asked_one_gpu.json"), ("bench_8ranks_rccl_asked.json", "bench_line_8ranks_rccl_asked_one_gpu.json"),
                 ("wideband.json", "wideband_f64_legs.json"), ("aux_rates.jsonl", "aux_rates.jsonl"), ("sliding_scan.jsonl", "sliding_scan.jsonl"), ("caf_rate.jsonl", "caf_rate.jsonl"),
                 ("tracked_rate.jsonl", "tracked_rate.jsonl"), ("n70_rate.txt", "n70_rate.txt")):
    shutil.copy(O + src, P + f"{tag}_" + dst)
import check_design
sys.exit(check_design.check(verbose=False))
