#!/bin/bash
# round 4, second GPU call: k_sums as two kernels, multi tests incl. MEX ngpu + bench --single-process, batch-1 sweep
out=gpurun_out/r04b; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_multi.py -m gpu -q -x --durations=8 > $out/pytest_multi.log 2>&1
echo "pytest rc $?" >> $out/pytest_multi.log
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --durations=10 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
: > $out/sweep.txt
for w in 2 4 8 16; do for sb in "1 1" "1 8" "3 8"; do set -- $sb
  r=$(TWX_SUMS_WGS=$w TWX_STREAMS=$1 python bench.py --steps 10 --warmup 2 --windows 192 --batch $2 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "TWX_SUMS_WGS=$w streams $1 batch $2 : $r" >> $out/sweep.txt
done; done
for b in 1 8; do
python bench.py --steps 5 --warmup 2 --windows 96 --batch $b --no-cpu-baseline --no-caf --no-pmc 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', json.dumps(j['kernels']))" >> $out/sweep.txt
done
python tools/aux_rates.py acq > $out/aux_rates.jsonl 2> $out/aux_rates.err
python bench.py --steps 10 --warmup 2 --cpu-windows 2 > $out/bench.json 2> $out/bench.err
tail -5 $out/pytest_multi.log; tail -8 $out/pytest.log; cat $out/sweep.txt; cat $out/aux_rates.jsonl | cut -c1-300; tail -c 1500 $out/bench.json
