cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_vitesse.py -q -x > $O/pytest_vitesse.log 2>&1; echo "rc $?" >> $O/pytest_vitesse.log
tail -25 $O/pytest_vitesse.log
