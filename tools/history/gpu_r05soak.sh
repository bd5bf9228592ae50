#!/bin/bash
# round 5: the randomised differential sweeps with large counts (a hunt, not part of the suite's default run)
# TWX_SWEEP_SEED=n shifts every sweep's generator seed (another draw of everything)
out=gpurun_out/r05soak; mkdir -p $out; : > $out/soak.txt
run() { echo "== TWX_SWEEP_SEED=${TWX_SWEEP_SEED:-0} TWX_SWEEP_OPTIONS=$1 $2 -k $3" >> $out/soak.txt; ( time TWX_SWEEP_OPTIONS=$1 timeout 2400 python -m pytest $2 -q -x -k "$3" ) 2>&1 | tail -8 | grep -v "^$" >> $out/soak.txt; }
run 3000 tests/test_gpu_parity.py "test_randomised_option_sweep"
run 2000 tests/test_gpu_parity.py "randomised_fir_and_sliding"
TWX_SLIDING_MFMA=1 run 1500 tests/test_gpu_parity.py "randomised_fir_and_sliding"
run 1000 tests/test_gpu_parity.py "randomised_caf_ranges"
run 1000 tests/test_gpu_parity.py "randomised_tracked_flows"
run 200 tests/test_gpu_multi.py "randomised_partitions"
run 300 tests/test_gpu_configs.py "randomised_acquisition"
run 100 tests/test_gpu_rx.py "randomised_receiver"
cat $out/soak.txt
