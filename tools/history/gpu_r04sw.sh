#!/bin/bash
# round 4: randomised parity sweeps on the final build (short codes: 9 000 windows; full window length: 264 windows of N = 5e6)
out=gpurun_out/r04sw; mkdir -p $out
( time TWX_SWEEP_WINDOWS=3000 timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_parity_sweep ) > $out/sweep.log 2>&1
timeout 2400 python tools/sweep_5m.py 24 > $out/sweep_5m.jsonl 2> $out/sweep_5m.err
timeout 3000 python tools/sweep_5m.py 240 41004 >> $out/sweep_5m.jsonl 2>> $out/sweep_5m.err
tail -6 $out/sweep.log; cat $out/sweep_5m.jsonl | cut -c1-600
