#!/bin/bash
out=gpurun_out/r03dma; mkdir -p $out
TWX_COL_PF=768 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
bash tools/history/gpu_ab.sh r03dma/ab "TWX_COL_PF=0" "TWX_COL_PF=768" "TWX_COL_PF=1536" "TWX_COL_PF=3072" "TWX_COL_PF=0" "TWX_COL_PF=768" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/ab/ab.txt
