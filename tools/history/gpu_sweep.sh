#!/bin/bash
# batch-size x pipeline-slot sweep (Infinity-Cache residency of the intermediates): tools/history/gpu_sweep.sh OUTDIR
out=gpurun_out/$1; mkdir -p $out
for s in 1 2 3; do for b in 1 2 4 8 16; do
  r=$(TWX_STREAMS=$s python bench.py --steps 10 --warmup 2 --windows 192 --batch $b --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "streams $s batch $b : $r" | tee -a $out/sweep.txt
done; done
