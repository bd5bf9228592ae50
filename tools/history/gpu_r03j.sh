#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or randomised or golden or two_channels or short_final" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
for pf in 512; do
  echo "TWX_ROW_PF=$pf: $(TWX_ROW_PF=$pf python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
done
bash tools/history/gpu_ab.sh r03j/ab "TWX_COL_PF=0" "TWX_COL_PF=1024" "TWX_COL_PF=0" "TWX_COL_PF=1024" "TWX_COL_PF=768" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/alone.txt; cat $out/ab/ab.txt
for pf in 0 1024 0 1024 768 2048; do
  echo "TWX_COL_PF=$pf: $(TWX_COL_PF=$pf python tools/kernel_alone.py k_col_inv 3 2>/dev/null | tail -1)" >> $out/alone_inv.txt
done
cat $out/alone_inv.txt
