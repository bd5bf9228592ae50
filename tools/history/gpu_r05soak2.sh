#!/bin/bash
# round 5, final build (fp64 column tiles of width 8, wideband session): the differential sweeps that touch the changed code, another seed
out=gpurun_out/r05soak2; mkdir -p $out; : > $out/soak.txt
run() { echo "== TWX_SWEEP_SEED=${TWX_SWEEP_SEED:-0} TWX_SWEEP_OPTIONS=$1 $2 -k $3" >> $out/soak.txt; ( time TWX_SWEEP_OPTIONS=$1 timeout 2400 python -m pytest $2 -q -x -k "$3" ) 2>&1 | tail -8 | grep -v "^$" >> $out/soak.txt; }
run 3000 tests/test_gpu_parity.py "test_randomised_option_sweep"
run 1000 tests/test_gpu_parity.py "randomised_fir_and_sliding"
run 600 tests/test_gpu_parity.py "randomised_caf_ranges"
run 600 tests/test_gpu_parity.py "randomised_tracked_flows"
run 200 tests/test_gpu_multi.py "randomised_partitions"
cat $out/soak.txt
