#!/bin/bash
out=gpurun_out/r03b16; mkdir -p $out
for b in 8 16 12 8 16; do
  r=$(python bench.py --steps 10 --warmup 2 --windows 192 --batch $b --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "batch $b : $r" | tee -a $out/sweep.txt
done
