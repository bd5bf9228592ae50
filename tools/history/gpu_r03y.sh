#!/bin/bash
out=gpurun_out/r03y; mkdir -p $out
for m in 1 2 3; do for s in 0 2 4 6; do
  echo "mode $m stagger $s: $(TWX_ROW_STAGGER=$s TWX_ROW_STAGGER_MODE=$m python tools/kernel_alone.py k_row_mid 2.5 2>/dev/null | tail -1)" >> $out/alone.txt
done; done
cat $out/alone.txt
