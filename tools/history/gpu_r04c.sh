#!/bin/bash
# round 4, third GPU call: the receiver program (twx_rx_*) against the oracle loop
out=gpurun_out/r04c; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_rx.py -m gpu -q -x --durations=8 > $out/pytest_rx.log 2>&1
echo "pytest rc $?" >> $out/pytest_rx.log
timeout 1200 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "rxcomplex or track_epoch or aux_kernels" > $out/pytest_aux.log 2>&1
echo "pytest rc $?" >> $out/pytest_aux.log
tail -40 $out/pytest_rx.log; tail -5 $out/pytest_aux.log
