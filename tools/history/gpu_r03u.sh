#!/bin/bash
out=gpurun_out/r03u; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
V=amaranth_twstft_amd/variants
bash tools/history/gpu_ab.sh r03u/ab "TWX_COL_PF=0" "TWX_X=1" "TWX_COL_PF=512" "TWX_COL_PF=1536" "TWX_SQ_FWD3=1" "TWX_SQ_FWD3=1 TWX_COL_PF=512" "TWX_COL_PF=0" "TWX_X=1" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/ab/ab.txt
