#!/bin/bash
# board power / shader clock under sustained synthetic loads (tools/power_probe.hip): tools/history/gpu_power2.sh OUTDIR
out=gpurun_out/$1; mkdir -p $out
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' '; echo; }
for m in read copy valu pkvalu lds "mix 5 1" "mix 5 4" "mix 5 16"; do
  tools/bin/power_probe $m 5 > $out/run.txt 2>&1 &
  pid=$!
  sleep 2.5; a=$(smi); sleep 1; b=$(smi)
  wait $pid
  echo "$m | $(cat $out/run.txt) | $a | $b" | tee -a $out/power.txt
  sleep 2
done
