#!/bin/bash
out=gpurun_out/r03sw; mkdir -p $out
( time TWX_SWEEP_WINDOWS=3000 timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_parity_sweep ) > $out/sweep.log 2>&1
tail -6 $out/sweep.log
