#!/bin/bash
out=gpurun_out/r03f7; mkdir -p $out
V=amaranth_twstft_amd/variants
bash tools/history/gpu_ab.sh r03f7/ab "TWX_X=1" "TWX_LIB=$V/lib_fwd3w7.so" "TWX_X=1" "TWX_LIB=$V/lib_fwd3w7.so" > /dev/null 2>&1
cat $out/ab/ab.txt
