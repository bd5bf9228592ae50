#!/bin/bash
# round 5: which job of k_sliding_mfma sets its time?  TWX_SM_ABLATE=1: no MFMAs, =2: no mixing (barriers only)
out=gpurun_out/${1:-r05m}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export TWX_SLIDING_MFMA=1
for a in 0 1 2; do
  export TWX_SM_ABLATE=$a
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$a -o sl -- python3 tools/aux_rates.py sliding_scan > /dev/null 2>&1
  f=$(find $out/prof$a -name "*kernel_stats.csv" | head -1)
  echo "ablate=$a"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    if 'sliding' in r['Name']: print('  ', r['Name'][40:110], r['Calls'], r['AverageNs'])
PY
done
