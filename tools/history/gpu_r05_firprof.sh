cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/fp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python3 tools/aux_rates.py > /tmp/fp.log 2>&1
f=$(find /tmp/fp -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
grep "k_fir" $f | sed "s/(HIP_vector_type.*)\",/\",/" | cut -c1-200
grep '"kernel": "k_fir' /tmp/fp.log | cut -c1-400
