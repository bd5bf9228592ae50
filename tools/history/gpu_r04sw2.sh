#!/bin/bash
# round 4: the documented A/B switches still give right answers (each with the tests that reach the code it switches)
run() { echo "== $1 : -k $2"; env $1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "$2" 2>&1 | grep -E "passed|failed|Error" | tail -2; }
run TWX_NO_DEINT=1 "all_channels or two_channels or option_sweep"
run TWX_CAF_SERIAL=1 "caf"
run TWX_CAF_ROTATE=0 "caf"
run TWX_SLIDING_NARROW=0 "sliding"
run TWX_SLIDING_WIDE8=0 "sliding"
run TWX_FIR_K=4 "fir or wideband or config4"
run TWX_SUMS_WGS=2 "processing_vs_oracle or option_sweep or batch_size"
run TWX_ROW_PF=0 "full_size or config3 or whole_correlation"
run TWX_ROW_PF=256 "full_size or whole_correlation"
run TWX_IO_THREADS=1 "process_file or tracked_file or ingest"
run TWX_TRK_BATCH=7 "tracked"
