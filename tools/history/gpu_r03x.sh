#!/bin/bash
# final build: sustained 400-step bench, k_rowd<MID> alone with power/clock samples, leak check
out=gpurun_out/r03x; mkdir -p $out
python3 bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-caf --no-pmc > $out/bench_sustained.json 2> $out/bench_sustained.err
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' '; echo; }
for k in k_row_mid; do
  python tools/kernel_alone.py $k 7 > $out/alone_$k.txt 2>&1 &
  pid=$!
  ( while kill -0 $pid 2>/dev/null; do echo "$(date +%s.%N | cut -c1-14) $(smi)"; sleep 0.5; done ) > $out/smi_$k.txt
  wait $pid
done
python3 tools/leak_check.py > $out/leak.txt 2>&1
python3 -c "
import json; j=json.loads(open('$out/bench_sustained.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['steps'], j['roofline']['frac'])"
cat $out/alone_k_row_mid.txt | tail -1; tail -6 $out/smi_k_row_mid.txt; tail -2 $out/leak.txt
