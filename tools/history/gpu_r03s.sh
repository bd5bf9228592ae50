#!/bin/bash
out=gpurun_out/r03s; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "fir or aux_kernels or wideband or 70msps or config4" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
python3 tools/aux_rates.py fir > $out/fir.jsonl 2>&1
tail -3 $out/pytest.log; cat $out/fir.jsonl
