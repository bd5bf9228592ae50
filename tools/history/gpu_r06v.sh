cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "two_forms" 2>&1 | tail -15 | tee $O/pytest.txt
TWX_SWEEP_SEED=5 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "random_bands" 2>&1 | tail -5 | tee -a $O/pytest.txt
