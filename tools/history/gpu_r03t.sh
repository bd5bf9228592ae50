#!/bin/bash
out=gpurun_out/r03t; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x -k "fir or aux_kernels or config4" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
python3 tools/aux_rates.py fir > $out/fir.jsonl 2>&1
V=amaranth_twstft_amd/variants
bash tools/history/gpu_ab.sh r03t/ab "TWX_X=1" "TWX_LIB=$V/lib_ablf1.so" "TWX_LIB=$V/lib_ablf3.so" "TWX_LIB=$V/lib_ablf4.so" "TWX_LIB=$V/lib_ablf5.so" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/fir.jsonl; cat $out/ab/ab.txt
