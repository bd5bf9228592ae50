#!/bin/bash
# round 4, sixth GPU call: squared-spectrum kernels rewritten (tracked flow), streaming sliding dot, k_df_tables in slices
out=gpurun_out/r04f; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sliding or tracked or sqspec or search_df" --durations=6 > $out/pytest_a.log 2>&1
echo "pytest rc $?" >> $out/pytest_a.log
timeout 900 python tools/tracked_rate.py 180 > $out/tracked_rate.jsonl 2> $out/tracked_rate.err
python tools/aux_rates.py sliding_scan > $out/sliding_scan.jsonl 2> $out/sliding_scan.err
TWX_SLIDING_NARROW=0 python tools/aux_rates.py sliding_scan 2>/dev/null | head -4 > $out/sliding_scan_general.jsonl
for sb in "1 1" "3 1" "3 8"; do set -- $sb
  r=$(TWX_STREAMS=$1 python bench.py --steps 10 --warmup 2 --windows 192 --batch $2 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
  echo "streams $1 batch $2 : $r" >> $out/sweep.txt
done
python tools/aux_rates.py acq 2>/dev/null | head -1 | cut -c1-300 > $out/interp.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest_a.log; cat $out/tracked_rate.jsonl; cat $out/sliding_scan.jsonl; echo general; cat $out/sliding_scan_general.jsonl; cat $out/sweep.txt $out/interp.txt; tail -5 $out/pytest.log
