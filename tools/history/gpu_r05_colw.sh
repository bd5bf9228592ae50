#!/bin/bash
# round 5: fp64 column passes with narrower tiles (W = 8: still 128-byte pieces of complex double, half the LDS: two workgroups per CU).
# first run (TWX_COL_W forcing built plug-ins): profiles/r05_f64_colw.txt; now the last pass of fp64 contexts takes W = 8 by itself
# (choose_col_inv, csrc/twx_api.hip) and TWX_COL_W=16 / 8 are the A/B legs (they force every column pass).
out=gpurun_out/r05colw; mkdir -p $out
leg() { python bench.py --wideband-only --wideband-seconds 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['f64_workload']; w=j['wideband_workload']; print('$1', 'f64', f['correlated_Msamples_per_s'], {k:v['avg_ms'] for k,v in f['kernels'].items()}, f['within_tolerance'], 'f32 corr', w['correlations_alone_Msamples_per_s'])"; }
leg "default (fp64: forward W=16, last pass W=8; fp32 W=16)" | tee -a $out/colw.txt
TWX_COL_W=16 leg "all W=16" | tee -a $out/colw.txt
TWX_COL_W=8 leg "all W=8" | tee -a $out/colw.txt
leg "default again" | tee -a $out/colw.txt
python -m pytest tests -x -q -m gpu -k "f64 or double or fp64" 2>&1 | tail -3 | tee -a $out/colw.txt
