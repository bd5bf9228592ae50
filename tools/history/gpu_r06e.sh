# round 6, GPU call E: k_col_inv3 with its last-stage twiddles parked in LDS (variant library) against the shipped form.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06e
rm -rf $O; mkdir -p $O
V=$R/amaranth_twstft_amd/variants/lib_twlds.so
for rep in 1 2; do
  echo "== shipped" >> $O/kernel_alone.txt; python3 tools/kernel_alone.py k_col_inv 2 2>/dev/null | tail -1 >> $O/kernel_alone.txt
  echo "== twiddles in LDS" >> $O/kernel_alone.txt; TWX_LIB=$V python3 tools/kernel_alone.py k_col_inv 2 2>/dev/null | tail -1 >> $O/kernel_alone.txt
done
cat $O/kernel_alone.txt
for v in shipped twlds shipped twlds; do
  if [ $v = twlds ]; then export TWX_LIB=$V; else unset TWX_LIB; fi
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_$v.json 2>/dev/null
  python3 - "$v" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/r06e/bench_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], j["value"], j["ms_per_step"], j["kernels"]["k_col_inv"], j["integer_lag_exact"], j["other_workload"]["value"])
PY
done
