#!/bin/bash
# round 5, review item 8: the file-fed rate — pread into the pinned slots (round 4) against memcpy from a mapping of the capture,
# 8 / 16 / 32 reader threads, plus the fp64 chain on the 1250 x 4000 pair (review item 7b: rows of 4000 leave room for two workgroups)
out=gpurun_out/r05io; mkdir -p $out
for m in 0 1; do
  echo "## TWX_FILE_MMAP=$m" >> $out/io.txt
  TWX_FILE_MMAP=$m python tools/io_rate.py 192 8 16 32 2>/dev/null >> $out/io.txt
done
cat $out/io.txt
echo "## fp64 chain, built-in pair 625 x 8000" > $out/f64.txt
python bench.py --wideband-only --wideband-seconds 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['f64_workload']; print(f['correlated_Msamples_per_s'], f['roofline']['frac'], {k:v['avg_ms'] for k,v in f['kernels'].items()})" >> $out/f64.txt
echo "## fp64 chain, TWX_N2=4000 (1250 x 4000, column tile W = 4)" >> $out/f64.txt
TWX_N2=4000 python bench.py --wideband-only --wideband-seconds 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['f64_workload']; w=j['wideband_workload']; print(f['correlated_Msamples_per_s'], f['roofline']['frac'], {k:v['avg_ms'] for k,v in f['kernels'].items()}, 'fp32 corr alone', w['correlations_alone_Msamples_per_s'])" >> $out/f64.txt
cat $out/f64.txt
