#!/bin/bash
# round 5: the sliding dot product on the matrix cores (k_sliding_mfma) against the packed-FMA form, parity first
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_rx.py -m gpu -x -q -k "sliding or track or rx or receiver" > gpurun_out/r05c/parity.log 2>&1
tail -4 gpurun_out/r05c/parity.log
TWX_SLIDING_MFMA=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliding" > gpurun_out/r05c/parity_forced.log 2>&1
tail -4 gpurun_out/r05c/parity_forced.log
for m in 0 1; do
  echo "TWX_SLIDING_MFMA=$m" >> gpurun_out/r05c/scan.txt
  TWX_SLIDING_MFMA=$m python tools/aux_rates.py sliding_scan >> gpurun_out/r05c/scan.txt 2>&1
done
cat gpurun_out/r05c/scan.txt
