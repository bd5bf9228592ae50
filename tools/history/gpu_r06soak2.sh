#!/bin/bash
# round 6: the randomised differential sweeps with large counts on the final build (new column passes, resample / extras / self-check options off and on)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06soak2; mkdir -p $out; : > $out/soak.txt
run() { echo "== TWX_SWEEP_SEED=${TWX_SWEEP_SEED:-0} TWX_SELFCHECK=${TWX_SELFCHECK:-0} TWX_SWEEP_OPTIONS=$1 $2 -k $3" >> $out/soak.txt; ( time TWX_SWEEP_OPTIONS=$1 timeout 1500 python3 -m pytest $2 -q -x -k "$3" ) 2>&1 | tail -8 | grep -v "^$" >> $out/soak.txt; }
export TWX_SWEEP_SEED=9
run 1500 tests/test_gpu_parity.py "test_randomised_option_sweep"
run 600 tests/test_gpu_parity.py "randomised_caf_ranges"
run 600 tests/test_gpu_parity.py "randomised_tracked_flows"
run 150 tests/test_gpu_multi.py "randomised_partitions"
TWX_SELFCHECK=1 run 600 tests/test_gpu_parity.py "test_randomised_option_sweep"
python3 tools/sweep_5m.py 48 9 > $out/sweep_5m.jsonl 2>/dev/null; tail -2 $out/sweep_5m.jsonl >> $out/soak.txt
cat $out/soak.txt
