# round 6, GPU call R: the whole GPU suite on the build with k_rowd_bandsum, smoke, a driver-style bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
