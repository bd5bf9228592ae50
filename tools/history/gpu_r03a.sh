#!/bin/bash
# round 3, first GPU call: tracked-flow tests through the C entry, bench collective proof, k_row_mid alone with power/clock samples
out=gpurun_out/r03a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tracked or squared_spectrum" --durations=10 > $out/pytest_tracked.log 2>&1
echo "rc $?" >> $out/pytest_tracked.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "bench" --durations=10 > $out/pytest_bench.log 2>&1
echo "rc $?" >> $out/pytest_bench.log
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' '; echo; }
for k in k_row_mid k_col_inv k_col_fwd_mix; do
  python tools/kernel_alone.py $k 7 > $out/alone_$k.txt 2>&1 &
  pid=$!
  ( while kill -0 $pid 2>/dev/null; do echo "$(date +%s.%N | cut -c1-14) $(smi)"; sleep 0.5; done ) > $out/smi_$k.txt
  wait $pid
done
tail -3 $out/pytest_tracked.log $out/pytest_bench.log; cat $out/alone_*.txt; for k in k_row_mid k_col_inv k_col_fwd_mix; do echo $k; tail -8 $out/smi_$k.txt; done
