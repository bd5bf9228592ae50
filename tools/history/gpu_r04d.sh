#!/bin/bash
# round 4, fourth GPU call: Infinity-Cache residency of the intermediates (small batches x plain instead of non-temporal accesses),
# the bench line with the CAF's PMC traffic, the whole GPU suite
out=gpurun_out/r04d; mkdir -p $out
: > $out/mall.txt
for lib in default ntoff ntbz0; do
  if [ $lib = default ]; then unset TWX_LIB; else export TWX_LIB=amaranth_twstft_amd/variants/lib_$lib.so; fi
  for sb in "1 1" "2 1" "3 1" "1 2" "2 2" "3 2" "2 4" "3 8"; do set -- $sb
    r=$(TWX_STREAMS=$1 python bench.py --steps 10 --warmup 2 --windows 192 --batch $2 --no-cpu-baseline --no-roofline --no-caf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.0f %s' % (j['value'], j['other_workload']['value'], j['integer_lag_exact']))")
    echo "$lib streams $1 batch $2 : $r" >> $out/mall.txt
  done
  for b in 1 2; do
    python bench.py --steps 5 --warmup 2 --windows 96 --batch $b --no-cpu-baseline --no-caf --no-pmc 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib batch $b', {k: v['avg_ms'] for k, v in j['kernels'].items()})" >> $out/mall.txt
  done
done
unset TWX_LIB
python bench.py --steps 10 --warmup 2 --cpu-windows 2 > $out/bench.json 2> $out/bench.err
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
cat $out/mall.txt; tail -8 $out/pytest.log; python -c "
import json; j=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(j['value'], json.dumps(j['caf_workload'])[:3000])"
