#!/bin/bash
out=gpurun_out/r03h; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
python3 tools/side_rates.py tracked > $out/tracked.txt 2>&1
tail -14 $out/pytest.log; tail -2 $out/tracked.txt
