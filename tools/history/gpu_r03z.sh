#!/bin/bash
out=gpurun_out/r03z; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batch or generator or windows" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
tail -2 $out/smoke.txt; tail -3 $out/pytest.log
