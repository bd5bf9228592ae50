#!/bin/bash
# round 6: third soak on the final build — two more seeds, larger counts, the receiver / acquisition / FIR sweeps too
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06soak3; mkdir -p $out; : > $out/soak.txt
run() { echo "== TWX_SWEEP_SEED=$TWX_SWEEP_SEED TWX_SELFCHECK=${TWX_SELFCHECK:-0} TWX_SWEEP_OPTIONS=$1 $2 -k $3" >> $out/soak.txt; TWX_SWEEP_OPTIONS=$1 timeout 1500 python3 -m pytest $2 -q -x -k "$3" 2>&1 | grep -E "passed|failed|error|Error" | tail -3 >> $out/soak.txt; }
for seed in 10 11; do
export TWX_SWEEP_SEED=$seed
run 2500 tests/test_gpu_parity.py "test_randomised_option_sweep"
run 1500 tests/test_gpu_parity.py "randomised_fir_and_sliding"
run 800 tests/test_gpu_parity.py "randomised_caf_ranges"
run 500 tests/test_gpu_parity.py "randomised_tracked_flows"
run 200 tests/test_gpu_multi.py "randomised_partitions"
run 300 tests/test_gpu_configs.py "randomised_acquisition"
run 100 tests/test_gpu_rx.py "randomised_receiver"
run 1 tests/test_gpu_parity.py "random_bands"
done
timeout 300 python3 tools/selfcheck_soak.py 48 > $out/selfcheck_soak.jsonl 2>/dev/null; cat $out/selfcheck_soak.jsonl >> $out/soak.txt
cat $out/soak.txt
