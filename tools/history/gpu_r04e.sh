#!/bin/bash
# round 4, fifth GPU call: CAF with rotated bin order (A/B), tracked flow at 180 s with the stage breakdown, full GPU suite
out=gpurun_out/r04e; mkdir -p $out
: > $out/caf_rotate.txt
for r in 0 1; do
  echo "TWX_CAF_ROTATE=$r $(TWX_CAF_ROTATE=$r python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_rotate.txt
done
for cfg in "64 16" "64 64" "128 32" "128 64" "256 64"; do set -- $cfg
  echo "rotate BPL=$1 BPW=$2 $(TWX_CAF_BPL=$1 TWX_CAF_BPW=$2 python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_rotate.txt
done
timeout 900 python tools/tracked_rate.py 180 > $out/tracked_rate.jsonl 2> $out/tracked_rate.err
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
python bench.py --steps 10 --warmup 2 --cpu-windows 2 > $out/bench.json 2> $out/bench.err
cat $out/caf_rotate.txt; cat $out/tracked_rate.jsonl; tail -3 $out/tracked_rate.err; tail -6 $out/pytest.log; python -c "
import json; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(j['value'], json.dumps(j['caf_workload'])[:2500])"
