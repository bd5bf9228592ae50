#!/bin/bash
out=gpurun_out/r03r; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "sliding or aux_kernels or tracking or track" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
python3 tools/aux_rates.py sliding > $out/sliding.jsonl 2>&1
python3 tools/aux_rates.py sliding_scan > $out/scan.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/aux_rates.py sliding > $out/stats.log 2>&1
tail -3 $out/pytest.log; cat $out/sliding.jsonl $out/scan.jsonl; cat $out/stats/*/*kernel_stats.csv | head -5
