#!/bin/bash
# clocks / power while the bench runs (is the chain power-limited?): tools/history/gpu_power.sh OUTDIR
out=gpurun_out/$1; mkdir -p $out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use|Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > $out/smi.txt 2>&1 &
sleep 1
python bench.py --steps 150 --warmup 5 --windows 192 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f other %.0f' % (j['value'], j['other_workload']['value']))" | tee $out/bench.txt
wait
sed -n 1,40p $out/smi.txt
