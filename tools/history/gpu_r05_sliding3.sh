#!/bin/bash
# round 5: k_sliding_mfma iteration — parity (forced), the scan with the matrix-core form, per-kernel durations (rocprofv3 CSV)
out=gpurun_out/${1:-r05e}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TWX_SLIDING_MFMA=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliding" > $out/parity_forced.log 2>&1
tail -2 $out/parity_forced.log
TWX_SLIDING_MFMA=1 python tools/aux_rates.py sliding_scan 2>/dev/null > $out/scan_mfma.txt
cat $out/scan_mfma.txt | cut -c1-200
export TWX_SLIDING_MFMA=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/slprof -o sl -- python3 tools/aux_rates.py sliding_scan > /dev/null 2>&1
f=$(find /tmp/slprof -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]: print(r['Name'][:100], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
