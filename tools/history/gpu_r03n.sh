#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
python tools/stamps_rowd.py > $out/stamps.txt 2>&1
for pf in 384 512 640 768 1024 1280 2500 0; do
  echo "TWX_ROW_PF=$pf: $(TWX_ROW_PF=$pf python tools/kernel_alone.py k_row_mid 3 2>/dev/null | tail -1)" >> $out/alone.txt
done
cat $out/stamps.txt | tail -40; cat $out/alone.txt
