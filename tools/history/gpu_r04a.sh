#!/bin/bash
# round 4, first GPU call: the new multi-GPU driver, parity after the k_sums rewrite, CAF overlap A/B, batch-1 sweep
out=gpurun_out/r04a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_multi.py -m gpu -q -x --durations=8 > $out/pytest_multi.log 2>&1
echo "pytest rc $?" >> $out/pytest_multi.log
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x --durations=10 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
: > $out/caf_overlap.txt
for s in 1 0; do
  echo "TWX_CAF_SERIAL=$s $(TWX_CAF_SERIAL=$s python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_overlap.txt
done
for cfg in "32 32" "32 16" "128 32"; do set -- $cfg
  echo "overlap BPL=$1 BPW=$2 $(TWX_CAF_BPL=$1 TWX_CAF_BPW=$2 python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_overlap.txt
done
: > $out/sweep.txt
for s in 1 3; do for b in 1 2 8; do
  r=$(TWX_STREAMS=$s python bench.py --steps 10 --warmup 2 --windows 192 --batch $b --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "streams $s batch $b : $r" >> $out/sweep.txt
done; done
for w in 2 4 16; do
  r=$(TWX_SUMS_WGS=$w TWX_STREAMS=1 python bench.py --steps 10 --warmup 2 --windows 192 --batch 1 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "TWX_SUMS_WGS=$w streams 1 batch 1 : $r" >> $out/sweep.txt
done
python tools/aux_rates.py > $out/aux_rates.jsonl 2> $out/aux_rates.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -5 $out/pytest_multi.log; tail -5 $out/pytest.log; cat $out/caf_overlap.txt $out/sweep.txt; cat $out/aux_rates.jsonl | cut -c1-400; tail -c 2500 $out/bench.json
