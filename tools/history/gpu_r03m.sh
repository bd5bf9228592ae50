#!/bin/bash
# resident-workgroup k_rowd<MID>: parity, then A/B against the previous build (variants/lib_head.so)
out=gpurun_out/r03m; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
V=amaranth_twstft_amd/variants
for lib in $V/lib_head.so amaranth_twstft_amd/libtwstft_hip.so $V/lib_head.so amaranth_twstft_amd/libtwstft_hip.so; do
  echo "$lib: $(TWX_LIB=$lib python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
done
echo "TWX_ROW_PF=0 (one workgroup per row): $(TWX_ROW_PF=0 python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
echo "TWX_ROW_PF=256: $(TWX_ROW_PF=256 python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
echo "TWX_ROW_PF=768: $(TWX_ROW_PF=768 python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
bash tools/history/gpu_ab.sh r03m/ab "TWX_LIB=$V/lib_head.so" "TWX_X=1" "TWX_LIB=$V/lib_head.so" "TWX_X=1" > /dev/null 2>&1
tail -3 $out/pytest.log; cat $out/alone.txt; cat $out/ab/ab.txt
