#!/bin/bash
# round 4: small batches (2, 3, 4 windows per launch, one slot) against the row pass's workgroups per launch
out=gpurun_out/r04q; mkdir -p $out; : > $out/b234_pf.txt
for b in 2 3 4; do for pf in -1 512 1024 1536 2048; do
  if [ $pf -lt 0 ]; then e=""; else e="TWX_ROW_PF=$pf"; fi
  r=$(env $e TWX_STREAMS=1 python bench.py --steps 8 --warmup 2 --windows 96 --batch $b --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "batch $b TWX_ROW_PF=$pf, 1 slot : $r" >> $out/b234_pf.txt
done; done
cat $out/b234_pf.txt
