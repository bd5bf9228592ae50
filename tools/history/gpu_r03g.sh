#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "track_epoch or rxcomplex or sliding or fir or aux_kernels" --durations=5 > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log
python3 tools/aux_rates.py > $out/aux_rates.jsonl 2> $out/aux_rates.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats_aux -- python3 $GRAFT_REPO_ROOT/tools/aux_rates.py sliding fir > $GRAFT_REPO_ROOT/$out/stats_aux.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 $out/pytest.log; cat $out/aux_rates.jsonl; head -8 $out/stats_aux/*/*_kernel_stats.csv
