#!/bin/bash
# round 4: the tracked flow with one batch per chunk (default now) against the library's 16 windows per launch (TWX_TRK_BATCH=16)
out=gpurun_out/r04q; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -x -k "tracked" 2>&1 | tail -3
for b in 0 16 32; do
  echo "TWX_TRK_BATCH=$b" >> $out/trk_batch.txt
  TWX_TRK_BATCH=$b python tools/tracked_rate.py 180 2>/dev/null | cut -c1-420 >> $out/trk_batch.txt
done
cat $out/trk_batch.txt
