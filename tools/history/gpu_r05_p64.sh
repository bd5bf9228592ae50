#!/bin/bash
# round 5: the row-walking (resident-workgroup) form of k_rowd<MID> for complex double (variant build -DTWX_MID_PERSIST64=1): parity of the fp64
# tests, then the fp64 leg of the bench for several resident-grid sizes (TWX_ROW_PF = workgroups in the launch; one fits a CU)
out=gpurun_out/r05p64; mkdir -p $out
V=amaranth_twstft_amd/variants/lib_p64.so
TWX_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "f64 or config4 or precision" > $out/parity.log 2>&1; tail -3 $out/parity.log
leg() { python bench.py --wideband-only --wideband-seconds 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['f64_workload']; print('$1', f['correlated_Msamples_per_s'], f['roofline']['frac'], f['kernels']['k_row_mid']['avg_ms'], f['within_tolerance'])"; }
echo "product build" | tee -a $out/p64.txt; leg product | tee -a $out/p64.txt
for pf in 256 512 768 1280; do
  export TWX_LIB=$V TWX_ROW_PF=$pf
  leg "persist64 pf=$pf" | tee -a $out/p64.txt
done
