#!/bin/bash
# round 4: host-fed rates against the PCIe link, interpolation chain after the table skip, receiver tests
out=gpurun_out/r04j; mkdir -p $out
python tools/io_rate.py 192 4 8 16 > $out/io_rate.txt 2>&1
python tools/aux_rates.py acq 2>/dev/null | head -1 | cut -c1-300 > $out/interp.txt
timeout 1500 python -m pytest tests/test_gpu_rx.py tests/test_gpu_configs.py -m gpu -q -x -k "receiver or rxcomplex or acquisition" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
grep -v amdgpu.ids $out/io_rate.txt; cat $out/interp.txt; tail -3 $out/pytest.log
