#!/bin/bash
out=gpurun_out/r03i32; mkdir -p $out
bash tools/history/gpu_ab.sh r03i32/ab "TWX_X=1" "TWX_INV_W32=1" "TWX_X=1" "TWX_INV_W32=1" > /dev/null 2>&1
cat $out/ab/ab.txt
