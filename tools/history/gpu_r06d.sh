# round 6, GPU call D: k_col_inv3w (resident workgroups walking the tiles, the next tile loaded ahead) against k_col_inv3.   bash tools/gpu_r06d.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06d
rm -rf $O; mkdir -p $O
for w in 0 2 4 0 2; do
  echo "== TWX_INV_WALK=$w" >> $O/kernel_alone.txt
  TWX_INV_WALK=$w python3 tools/kernel_alone.py k_col_inv 2 2>/dev/null | tail -1 >> $O/kernel_alone.txt
done
cat $O/kernel_alone.txt
for w in 0 2 0 2; do
  TWX_INV_WALK=$w timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_walk$w.json 2>/dev/null
  python3 - "$w" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/r06d/bench_walk%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("walk", sys.argv[1], j["value"], j["ms_per_step"], j["kernels"]["k_col_inv"], j["integer_lag_exact"], j["other_workload"]["value"])
PY
done
TWX_INV_WALK=2 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity_walk2.log 2>&1; tail -3 $O/pytest_parity_walk2.log
