#!/bin/bash
# round-end rehearsal: smoke, the -m gpu suite, the driver-style bench line twice
out=gpurun_out/r03fin; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 2800 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
for i in 1 2; do ( time python3 bench.py > $out/bench_$i.json 2> $out/bench_$i.err ) 2> $out/time_$i.txt; done
tail -1 $out/smoke.txt; tail -3 $out/pytest.log
for i in 1 2; do python3 -c "
import json; j=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['avg_ms'], r['traffic'], r.get('traffic_source','')[:40], j['caf_workload']['s_per_window'])"; grep real $out/time_$i.txt; done
