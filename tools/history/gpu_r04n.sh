#!/bin/bash
# round 4: the eight-output FIR form against the four-output one (TWX_FIR_K=4)
out=gpurun_out/r04n; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -x -k "fir or wideband" 2>&1 | tail -5
python -m pytest tests/test_gpu_configs.py -q -x -k "config4" 2>&1 | tail -3
python tools/aux_rates.py fir > $out/fir_k8.jsonl 2>$out/err.txt
TWX_FIR_K=4 python tools/aux_rates.py fir > $out/fir_k4.jsonl 2>>$out/err.txt
cat $out/fir_k8.jsonl $out/fir_k4.jsonl | cut -c1-330
