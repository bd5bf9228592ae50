#!/bin/bash
# round 4, seventh GPU call: k_rowd<MID> with 16-byte Bz stores (A/B), kernel profile of the tracked flow
out=gpurun_out/r04g; mkdir -p $out
for lib in amaranth_twstft_amd/libtwstft_hip.so amaranth_twstft_amd/variants/lib_st16.so amaranth_twstft_amd/libtwstft_hip.so amaranth_twstft_amd/variants/lib_st16.so; do
  echo "$lib: $(TWX_LIB=$lib python tools/kernel_alone.py k_row_mid 3 | tail -1)" >> $out/st16_alone.txt
done
bash tools/history/gpu_ab.sh r04g_ab "TWX_X=0" "TWX_LIB=amaranth_twstft_amd/variants/lib_st16.so" "TWX_X=1" "TWX_LIB=amaranth_twstft_amd/variants/lib_st16.so" > /dev/null 2>&1
cp gpurun_out/r04g_ab/ab.txt $out/st16_ab.txt
TWX_LIB=amaranth_twstft_amd/variants/lib_st16.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or randomised or processing_vs_oracle" > $out/pytest_st16.log 2>&1
echo "pytest rc $?" >> $out/pytest_st16.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_trk -- python3 $GRAFT_REPO_ROOT/tools/tracked_rate.py 40 > $GRAFT_REPO_ROOT/$out/prof_trk.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/prof_trk -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-200 > $out/prof_trk_kernel_stats.txt
rm -rf $out/prof_trk
cat $out/st16_alone.txt $out/st16_ab.txt; tail -3 $out/pytest_st16.log; cat $out/prof_trk_kernel_stats.txt
