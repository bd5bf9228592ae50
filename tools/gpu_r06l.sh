cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_configs.py -q -x -k "code_stepping or early_prompt" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
