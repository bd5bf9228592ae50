#!/bin/bash
out=gpurun_out/r03w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "pmc_traffic or world_of_one" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
( time python3 bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/time.txt
tail -4 $out/pytest.log; cat $out/time.txt; python3 -c "
import json; j=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['roofline'])"
