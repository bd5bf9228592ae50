# round 6, GPU call B: the self-check (tests, soak beside the unfenced matrix-core FIR, cost), the probe of k_col_inv's read shape on
# random data.   bash tools/gpu_r06b.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06b
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_selfcheck.py -x -q -s > $O/pytest_selfcheck.log 2>&1; echo "pytest rc $?" >> $O/pytest_selfcheck.log
tail -15 $O/pytest_selfcheck.log
timeout 600 python3 tools/selfcheck_soak.py 48 > $O/selfcheck_soak.jsonl 2> $O/selfcheck_soak.err
cat $O/selfcheck_soak.jsonl
timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_selfcheck_off.json 2> $O/bench_off.err
TWX_SELFCHECK=1 timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_selfcheck_on.json 2> $O/bench_on.err
python3 - <<'PY'
import json
for f in ("off", "on"):
    try:
        j = json.loads([l for l in open("gpurun_out/r06b/bench_selfcheck_%s.json" % f) if l.startswith("{")][-1])
        print(f, j["value"], j["ms_per_step"], j["kernels"]["k_row_mid"], j["integer_lag_exact"])
    except Exception as e:
        print(f, "unreadable", e)
PY
tools/bin/bw_probe 2 > $O/bw_probe_constant.txt 2>&1
tools/bin/bw_probe 2 rand > $O/bw_probe_random.txt 2>&1
grep -i "colinv\|rowmid\|device" $O/bw_probe_constant.txt $O/bw_probe_random.txt
python3 tools/kernel_alone.py k_col_inv 2 > $O/kernel_alone_col_inv.txt 2>&1
python3 tools/kernel_alone.py k_col_inv 2 zero > $O/kernel_alone_col_inv_zero.txt 2>&1
tail -1 $O/kernel_alone_col_inv.txt $O/kernel_alone_col_inv_zero.txt
