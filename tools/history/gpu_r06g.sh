# round 6, GPU call G: the whole GPU suite on the new column passes, the driver-style line, the rocprof statistics.   bash tools/gpu_r06g.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06g
rm -rf $O; mkdir -p $O
P=/tmp/r06g; rm -rf $P; mkdir -p $P
keep() { f=$(find $P/$1 -name "*$2" | head -1); [ -n "$f" ] && cp "$f" $O/$1_$2; }
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
TWX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_1slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc --no-wideband > $O/stats_1slot.log 2>&1
keep stats_1slot kernel_stats.csv
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06g/bench_default.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["kernels"], j["other_workload"]["value"], j["caf_workload"]["s_per_window"], j["wideband_workload"]["input_Msamples_per_s"], j["f64_workload"]["correlated_Msamples_per_s"])
PY
