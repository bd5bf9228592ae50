# round 6, GPU call M: the file-fed rate with the persistent reader pool; the file / multi tests on it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06m
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  timeout 120 python3 tools/io_rate.py 96 8 16 32 2>/dev/null >> $O/io_rate.txt
done
cat $O/io_rate.txt
timeout 900 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -q -x -k "file or multi or partition or ingest or pipeline or repeated" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
