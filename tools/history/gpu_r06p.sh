# round 6, GPU call P: k_rowd_bandsum (the pruned row pass of the carrier search without the row in LDS) — parity, then A/B against k_rowd<BAND>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
for v in 0 1 0 1; do
  TWX_BANDSUM=$v timeout 120 python3 tools/kernel_alone.py k_row_band 3 2>&1 | tail -2 | sed "s/^/alone bandsum=$v /" | tee -a $O/ab.txt
done
for v in 0 1 0 1; do
  TWX_BANDSUM=$v timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_$v.json 2>$O/bench_$v.err
  python3 - "$v" <<'PY' | tee -a $O/ab.txt
import json, sys
j = json.loads([l for l in open("gpurun_out/r06p/bench_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("bench bandsum=%s" % sys.argv[1], j["value"], j["other_workload"]["value"], j["integer_lag_exact"], j.get("kernels", {}).get("k_row_band"), j["roofline"]["avg_ms"])
PY
done
