#!/bin/bash
# round 4: full GPU suite (incl. the 8-rank bench test) on the final build, then the profile set
out=gpurun_out/r04i; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -20 $out/pytest.log
bash tools/prof_round4.sh > $out/prof_round4.log 2>&1
tail -30 $out/prof_round4.log | cut -c1-600
