cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/caf; mkdir -p gpurun_out/caf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/caf/stats -- python3 tools/side_rates.py caf > gpurun_out/caf/log.txt 2>&1
tail -2 gpurun_out/caf/log.txt
