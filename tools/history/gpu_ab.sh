#!/bin/bash
# A/B of kernel variants selected by environment variables: tools/history/gpu_ab.sh OUTDIR "ENV1=.. ENV2=.." "ENV..." ...
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "=== variant $i: $envs" | tee -a $out/ab.txt
  env $envs python bench.py --steps 10 --warmup 2 --windows 192 --no-cpu-baseline --no-pmc > $out/bench_$i.json 2> $out/bench_$i.err
  python - "$out/bench_$i.json" <<'PY' | tee -a $out/ab.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.0f Msps  ms/step %.3f  lag_exact %s  other %.0f" % (j["value"], j["ms_per_step"], j["integer_lag_exact"], j["other_workload"]["value"]))
    print("  ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in j["kernels"].items()))
except Exception as e:
    print("FAILED", e)
PY
done
