cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\.\|^$" | tail -60 > $O/pytest.txt; head -c 6000 $O/pytest.txt
