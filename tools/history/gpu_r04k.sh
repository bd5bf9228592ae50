#!/bin/bash
# round 4: ingest with per-piece copies (twx_process_file / twx_process_windows), leak check incl. multi + receiver
out=gpurun_out/r04k; mkdir -p $out
python tools/io_rate.py 192 8 4 8 16 > $out/io_rate.txt 2>&1
TWX_STREAMS=4 python tools/io_rate.py 192 8 16 > $out/io_rate_4slots.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -m gpu -q -x -k "file or host or channel or multi or config3 or ranks or script or pipeline or mex" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
timeout 900 python tools/leak_check.py > $out/leak_check.txt 2>&1
grep -v amdgpu.ids $out/io_rate.txt; grep -v amdgpu.ids $out/io_rate_4slots.txt; tail -3 $out/pytest.log; tail -6 $out/leak_check.txt
