#!/bin/bash
# round 4: wide sliding windows with eight samples per lane against four (TWX_SLIDING_WIDE8=0)
out=gpurun_out/r04o; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -x -k "sliding" 2>&1 | tail -5
python -m pytest tests/test_gpu_rx.py tests/test_gpu_configs.py -q -x -k "receiver or rxcomplex or track" 2>&1 | tail -3
python tools/aux_rates.py sliding sliding_scan > $out/sliding_w8.jsonl 2>$out/err.txt
TWX_SLIDING_WIDE8=0 python tools/aux_rates.py sliding sliding_scan > $out/sliding_w4.jsonl 2>>$out/err.txt
grep "28\|16" $out/sliding_w8.jsonl | cut -c1-260; echo; grep "28\|16" $out/sliding_w4.jsonl | cut -c1-260
