#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
python tools/caf_rate.py > $out/caf_rate.jsonl 2>/dev/null
python bench.py > $out/bench.json 2> $out/bench.err
tail -25 $out/pytest.log; cat $out/caf_rate.jsonl; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03f/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline'], j['caf_workload'])
PY
