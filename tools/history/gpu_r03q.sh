#!/bin/bash
# full -m gpu suite on the final library, then the round-3 profile set
out=gpurun_out/r03q; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=5 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
bash tools/history/prof_round3.sh > $out/prof.log 2>&1
tail -12 $out/pytest.log; tail -c 1500 gpurun_out/r03p/bench_default.json
