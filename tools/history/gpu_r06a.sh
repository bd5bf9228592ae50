# round 6, GPU call A: the whole GPU suite on the new build (no spilling kernels, the fence, the sharded recording), the N = 1 line, the
# 8-rank rehearsal of the strong leg, and the cache counters of k_col_inv3 beside the probe of its read shape.   bash tools/gpu_r06a.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06a
rm -rf $O; mkdir -p $O
P=/tmp/r06a; rm -rf $P; mkdir -p $P
keep() { f=$(find $P/$1 -name "*$2" | head -1); [ -n "$f" ] && cp "$f" $O/$1_$2; }
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python3 bench.py --gpus 8 --steps 5 --warmup 2 --windows 600 --no-roofline > $O/bench_8ranks_600.json 2> $O/bench_8ranks_600.err
timeout 600 python3 bench.py --gpus 8 --single-process --steps 5 --warmup 2 --windows 600 > $O/bench_single_process_8ctx_600.json 2> $O/bench_single_process_8ctx.err
rocprofv3 -L > $O/counters_avail.txt 2>&1
for c in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCC_TAG_STALL_sum SQ_WAIT_INST_ANY SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/probe_$c -- tools/bin/bw_probe 2 > $O/probe_$c.log 2>&1
  f=$(find $P/probe_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -i "colinv\|Counter_Name" $f | head -40 > $O/probe_$c.csv
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/kern_$c -- python3 tools/kernel_alone.py k_col_inv 0.2 > $O/kern_$c.log 2>&1
  f=$(find $P/kern_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -i "k_col_inv\|Counter_Name" $f | head -12 > $O/kern_$c.csv
  case $c in TCC_HIT_sum|TCC_MISS_sum|TCP_PENDING_STALL_CYCLES_sum|TA_BUSY_avr)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/band_$c -- python3 tools/kernel_alone.py k_row_band 0.2 > $O/band_$c.log 2>&1
  f=$(find $P/band_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -i "k_rowd\|Counter_Name" $f | head -12 > $O/band_$c.csv ;;
  esac
done
tools/bin/bw_probe 2 > $O/bw_probe.txt 2>&1
python3 tools/kernel_alone.py k_col_inv 3 > $O/kernel_alone_col_inv.txt 2>&1
ls $O | wc -l
tail -c 1200 $O/bench_default.json; echo; python3 - <<'PY'
import json
for f in ("gpurun_out/r06a/bench_8ranks_600.json", "gpurun_out/r06a/bench_single_process_8ctx_600.json"):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, j["value"], j.get("strong_workload"))
    except Exception as e:
        print(f, "unreadable", e)
PY
