#!/bin/bash
# round 4, eighth GPU call: tracked flow with per-piece staging, pieces per chunk 4 / 8 / 16
out=gpurun_out/r04h; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "tracked or two_way or sqspec or search_df" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
for t in 4 8 16; do
  echo "TWX_IO_THREADS=$t" >> $out/tracked_rate.txt
  TWX_IO_THREADS=$t timeout 600 python tools/tracked_rate.py 180 2>/dev/null >> $out/tracked_rate.txt
done
tail -3 $out/pytest.log; cut -c1-420 $out/tracked_rate.txt
