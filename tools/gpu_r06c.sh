# round 6, GPU call C: the velocity-compensated window, the SNR estimators, the whole suite again.   bash tools/gpu_r06c.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06c
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_vitesse.py tests/test_gpu_snr_estimators.py -q -s > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log
tail -40 $O/pytest_new.log
echo skipped-full-suite
tail -6 $O/pytest_gpu.log
