#!/bin/bash
# is the dominant kernel power-limited?  the same instruction stream on an all-zero capture (no data toggling) against real data
out=gpurun_out/r03i; mkdir -p $out
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' '; echo; }
for mode in real zero real zero; do
  python tools/kernel_alone.py k_row_mid 5 $mode > $out/run.txt 2>&1 &
  pid=$!
  sleep 9; a=$(smi); sleep 1; b=$(smi)
  wait $pid
  echo "$(tail -1 $out/run.txt) | $a | $b" | tee -a $out/zero_vs_real.txt
done
