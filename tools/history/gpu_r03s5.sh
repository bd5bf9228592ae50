#!/bin/bash
out=gpurun_out/r03s5; mkdir -p $out
python3 tools/sweep_5m.py 24 > $out/sweep.jsonl 2> $out/err.txt
python3 tools/sweep_5m.py 240 31003 >> $out/sweep.jsonl 2>> $out/err.txt
cat $out/sweep.jsonl; tail -3 $out/err.txt
