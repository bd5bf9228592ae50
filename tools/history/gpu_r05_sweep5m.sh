#!/bin/bash
# round 5, final build: the full-length (N = 5e6) randomised parity sweep, 264 seeded windows (signal down to pure noise), fp32 and fp64 contexts
out=gpurun_out/r05s5; mkdir -p $out
timeout 2400 python tools/sweep_5m.py 24 > $out/sweep_5m.jsonl 2> $out/sweep_5m.err
timeout 3000 python tools/sweep_5m.py 240 51005 >> $out/sweep_5m.jsonl 2>> $out/sweep_5m.err
cut -c1-700 $out/sweep_5m.jsonl; tail -3 $out/sweep_5m.err
