#!/bin/bash
# round 5, review item 4: what would a compact (fp16) Bz buy?  Upper bound: the product build against a build whose fp32 contexts
# keep Bz as fp16 pairs (TWX_BZ16=1: k_rowd<MID> stores, k_col_inv3 / k_peak load 4 bytes per element), same bench, kernel times
# from the HIP-event profile of the bench itself.  The variant has no candidate / exact-mode pass: its lags are those of the fp16 map.
out=gpurun_out/r05bz; mkdir -p $out
for v in "" "amaranth_twstft_amd/variants/lib_bz16.so"; do
  echo "=== TWX_LIB=$v" | tee -a $out/ab.txt
  for rep in 1 2; do
    TWX_LIB=$v python bench.py --steps 10 --warmup 2 --windows 192 --no-cpu-baseline --no-pmc --no-caf --no-wideband > $out/b.json 2> $out/b.err
    python - $out/b.json <<'PY' | tee -a $out/ab.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.0f Msps  ms/step %.3f  lag_exact %s  other %.0f" % (j["value"], j["ms_per_step"], j["integer_lag_exact"], j["other_workload"]["value"]))
    print("  ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in j["kernels"].items()))
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
  done
done
