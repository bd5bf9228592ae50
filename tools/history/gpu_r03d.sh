#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or tracked_file or caf" --durations=8 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
: > $out/caf_sweep.txt
for cfg in "64 32 -1" "64 32 0" "16 16 -1" "16 8 -1" "8 8 -1" "8 4 -1" "8 2 -1" "4 4 -1" "4 2 -1" "4 1 -1" "2 2 -1" "2 1 -1" "4 4 1"; do
  set -- $cfg
  if [ "$3" = "-1" ]; then unset TWX_CAF_NT; else export TWX_CAF_NT=$3; fi
  echo "BPL=$1 BPW=$2 NT=$3 $(TWX_CAF_BPL=$1 TWX_CAF_BPW=$2 python tools/caf_rate.py 2>/dev/null | tail -1)" >> $out/caf_sweep.txt
done
unset TWX_CAF_NT
python bench.py --steps 10 --warmup 2 > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_aux -- python3 $GRAFT_REPO_ROOT/tools/aux_rates.py sliding fir > $GRAFT_REPO_ROOT/$out/prof_aux.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/prof_aux -name "*kernel_stats.csv" | head -1 | xargs -r head -12 > $out/prof_aux_kernel_stats.txt
rm -rf $out/prof_aux
tail -5 $out/pytest.log; cat $out/caf_sweep.txt; tail -c 3000 $out/bench.json; cat $out/prof_aux_kernel_stats.txt
