#!/bin/bash
# round 4: k_rowd<MID> workgroups per launch (TWX_ROW_PF) x pipeline slots in the CHAIN: does a row pass that leaves LDS free let the
# column passes of the other slots share its CUs?
out=gpurun_out/r04m; mkdir -p $out
: > $out/pf_slots.txt
for pf in 256 384 512 768 1280; do for s in 2 3 4; do
  r=$(TWX_ROW_PF=$pf TWX_STREAMS=$s python bench.py --steps 10 --warmup 2 --windows 192 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "TWX_ROW_PF=$pf streams $s : $r" >> $out/pf_slots.txt
done; done
cat $out/pf_slots.txt
