#!/bin/bash
# round-2 first GPU pass: bandwidth probe, the whole -m gpu suite, the default bench line
mkdir -p gpurun_out/r02a
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
tools/bin/bw_probe 2 > gpurun_out/r02a/bw_probe.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
timeout 900 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench rc $?" >> gpurun_out/r02a/bench.err
tail -3 gpurun_out/r02a/pytest.log; tail -c 1500 gpurun_out/r02a/bench.json
