# round 6, GPU call K: k_rowd<BAND> with resident workgroups (variant library) against the shipped form
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06k
rm -rf $O; mkdir -p $O
for v in shipped bandp shipped bandp; do
  if [ $v = shipped ]; then unset TWX_LIB; else export TWX_LIB=$R/amaranth_twstft_amd/variants/lib_$v.so; fi
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_$v.json 2>/dev/null
  python3 - "$v" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/r06k/bench_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
k = j["kernels"]
print(sys.argv[1], j["value"], j["ms_per_step"], "band", k["k_row_band"]["avg_ms"], "sq", k["k_col_fwd_square"]["avg_ms"], j["integer_lag_exact"])
PY
done
