#!/bin/bash
out=gpurun_out/r03o; mkdir -p $out
V=amaranth_twstft_amd/variants
for lib in $V/lib_nost.so $V/lib_noarith.so amaranth_twstft_amd/libtwstft_hip.so; do
  for pf in 512 1280; do
  echo "$lib PF=$pf: $(TWX_ROW_PF=$pf TWX_LIB=$lib python tools/kernel_alone.py k_row_mid 3 2>/dev/null | tail -1)" >> $out/alone.txt
  done
done
for pf in 896 1024 1152 1280 1536 1792; do
  echo "TWX_ROW_PF=$pf: $(TWX_ROW_PF=$pf python tools/kernel_alone.py k_row_mid 3 2>/dev/null | tail -1)" >> $out/alone.txt
done
bash tools/history/gpu_ab.sh r03o/ab "TWX_ROW_PF=512" "TWX_ROW_PF=1024" "TWX_ROW_PF=1280" "TWX_ROW_PF=512" "TWX_ROW_PF=1024" "TWX_ROW_PF=1280" > /dev/null 2>&1
cat $out/alone.txt; cat $out/ab/ab.txt
