#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x -k "rxcomplex or aux_kernels or sliding or fir or tracked_file or two_way or tracking" --durations=8 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
python tools/aux_rates.py > $out/aux_rates.jsonl 2> $out/aux_rates.err
tail -12 $out/pytest.log; cat $out/aux_rates.jsonl; tail -3 $out/aux_rates.err
