#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $out/pytest_parity.log 2>&1
echo "pytest rc $?" >> $out/pytest_parity.log
V=amaranth_twstft_amd/variants
bash tools/history/gpu_ab.sh r03e/ab "TWX_LIB=$V/lib_base.so" "TWX_LIB=$V/lib_fold.so" "TWX_LIB=$V/lib_sgpr.so" "TWX_X=1" "TWX_LIB=$V/lib_base.so" "TWX_X=1" > /dev/null 2>&1
for k in k_row_mid; do
  for lib in $V/lib_base.so amaranth_twstft_amd/libtwstft_hip.so; do
    echo "$lib: $(TWX_LIB=$lib python tools/kernel_alone.py $k 4 2>/dev/null | tail -1)" >> $out/alone.txt
  done
done
: > $out/caf_sweep2.txt
for cfg in "64 32 2560" "128 32 6000" "256 32 12000" "256 64 12000" "512 32 24000"; do
  set -- $cfg
  echo "BPL=$1 BPW=$2 MAXMB=$3 $(TWX_CAF_BPL=$1 TWX_CAF_BPW=$2 TWX_CAF_MAXMB=$3 python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_sweep2.txt
done
tail -4 $out/pytest_parity.log; cat $out/ab/ab.txt; cat $out/alone.txt; cat $out/caf_sweep2.txt
