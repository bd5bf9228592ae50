#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
V=amaranth_twstft_amd/variants
for lib in $V/lib_nogen.so amaranth_twstft_amd/libtwstft_hip.so $V/lib_nogen.so amaranth_twstft_amd/libtwstft_hip.so; do
  echo "$lib: $(TWX_LIB=$lib python tools/kernel_alone.py k_row_mid 4 2>/dev/null | tail -1)" >> $out/alone.txt
  echo "$lib: $(TWX_LIB=$lib python tools/kernel_alone.py k_row_band 3 2>/dev/null | tail -1)" >> $out/alone.txt
done
bash tools/history/gpu_ab.sh r03k/ab "TWX_LIB=$V/lib_nogen.so" "TWX_X=1" "TWX_LIB=$V/lib_nogen.so" "TWX_X=1" > /dev/null 2>&1
python tools/stamps_rowd.py > $out/stamps.txt 2>&1
tail -3 $out/pytest.log; cat $out/alone.txt; cat $out/ab/ab.txt; tail -32 $out/stamps.txt
