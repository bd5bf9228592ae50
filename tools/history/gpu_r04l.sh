#!/bin/bash
# round 4: CAF with ONE row-pass workgroup per CU (unused dynamic LDS) beside the other stream's column pass; 8 contexts from one process; 8 gloo ranks
out=gpurun_out/r04l; mkdir -p $out
: > $out/caf_pad.txt
for pad in 0 5120 6144 6656 45056; do
  echo "TWX_CAF_PAD=$pad $(TWX_CAF_PAD=$pad python tools/caf_rate.py 2>&1 | tail -1 | cut -c1-200)" >> $out/caf_pad.txt
done
for cfg in "6144 64 16" "6144 32 16" "6144 32 32" "6144 128 32"; do set -- $cfg
  echo "TWX_CAF_PAD=$1 BPL=$2 BPW=$3 $(TWX_CAF_PAD=$1 TWX_CAF_BPL=$2 TWX_CAF_BPW=$3 python tools/caf_rate.py 2>&1 | tail -1 | cut -c1-200)" >> $out/caf_pad.txt
done
python bench.py --gpus 8 --single-process --steps 5 --warmup 2 --windows 75 > $out/bench_single_process_8ctx.json 2> $out/bench_sp8.err
python bench.py --gpus 8 --backend gloo --steps 5 --warmup 2 --windows 75 --no-cpu-baseline --no-roofline > $out/bench_8ranks_gloo.json 2> $out/bench_g8.err
cat $out/caf_pad.txt; tail -c 900 $out/bench_single_process_8ctx.json; echo; tail -c 900 $out/bench_8ranks_gloo.json
