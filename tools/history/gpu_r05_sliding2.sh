#!/bin/bash
# round 5: k_sliding_mfma per-kernel durations (rocprofv3 --kernel-trace --stats) and the scan, both forms
mkdir -p gpurun_out/r05d
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TWX_SLIDING_MFMA=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliding" > gpurun_out/r05d/parity_forced.log 2>&1
tail -2 gpurun_out/r05d/parity_forced.log
for m in 0 1; do
  echo "TWX_SLIDING_MFMA=$m" >> gpurun_out/r05d/scan.txt
  TWX_SLIDING_MFMA=$m python tools/aux_rates.py sliding_scan 2>/dev/null >> gpurun_out/r05d/scan.txt
done
cat gpurun_out/r05d/scan.txt
export TWX_SLIDING_MFMA=1
rocprofv3 --kernel-trace --stats -d gpurun_out/r05d/prof -o sl -- python3 tools/aux_rates.py sliding_scan > /dev/null 2>&1
f=$(ls gpurun_out/r05d/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find gpurun_out/r05d/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-260
