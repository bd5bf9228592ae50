#!/bin/bash
# round 4: the one-window (batch 1, one slot) chain against the row pass's workgroups per launch: 625 rows on 512 resident slots
out=gpurun_out/r04q; mkdir -p $out; : > $out/b1_pf.txt
for pf in -1 0 256 320 512 625 640; do
  if [ $pf -lt 0 ]; then e=""; else e="TWX_ROW_PF=$pf"; fi
  r=$(env $e TWX_STREAMS=1 python bench.py --steps 10 --warmup 2 --windows 96 --batch 1 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "TWX_ROW_PF=$pf batch 1, 1 slot : $r" >> $out/b1_pf.txt
done
cat $out/b1_pf.txt
