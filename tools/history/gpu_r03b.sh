#!/bin/bash
# round 3: the whole -m gpu suite after the tracked-flow move
out=gpurun_out/r03b; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -40 $out/pytest.log
