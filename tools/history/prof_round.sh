cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/r01f; mkdir -p gpurun_out/r01f
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r01f/bench_default.json 2> gpurun_out/r01f/bench_default.err
TWX_STREAMS=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01f/bench_1slot.json 2>/dev/null
TWX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/stats_1slot -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01f/stats_1slot.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/stats_3slot -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01f/stats_3slot.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01f/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > gpurun_out/r01f/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01f/pmc_write -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > gpurun_out/r01f/pmc_write.log 2>&1
find gpurun_out/r01f -name "*.csv" | head -20
tail -c 600 gpurun_out/r01f/bench_default.json
