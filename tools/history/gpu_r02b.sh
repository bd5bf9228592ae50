#!/bin/bash
mkdir -p gpurun_out/r02b
tools/bin/valu_probe > gpurun_out/r02b/valu_probe.txt 2>&1
tools/bin/bw_probe 2 > gpurun_out/r02b/bw_probe.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r02b/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02b/pytest.log
cat gpurun_out/r02b/valu_probe.txt; tail -25 gpurun_out/r02b/bw_probe.txt; tail -5 gpurun_out/r02b/pytest.log
