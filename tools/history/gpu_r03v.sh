#!/bin/bash
out=gpurun_out/r03v; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
V=amaranth_twstft_amd/variants
bash tools/history/gpu_ab.sh r03v/ab "TWX_LIB=$V/lib_head.so" "TWX_X=1" "TWX_LIB=$V/lib_head.so" "TWX_X=1" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/ab/ab.txt
