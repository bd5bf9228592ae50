# round 6, GPU call F: the shipped library (k_col_inv3 twiddles in LDS) against variants with the same in k_col_fwd3 (MIX) and for SQUARE too
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06f
rm -rf $O; mkdir -p $O
for v in shipped sq3 mix1 mix2 sq1 sq2 shipped sq3 mix1 mix2; do
  if [ $v = shipped ]; then unset TWX_LIB; else export TWX_LIB=$R/amaranth_twstft_amd/variants/lib_$v.so; fi
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_$v.json 2>/dev/null
  python3 - "$v" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/r06f/bench_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
k = j["kernels"]
print(sys.argv[1], j["value"], j["ms_per_step"], "sq", k["k_col_fwd_square"]["avg_ms"], "mix", k["k_col_fwd_mix"]["avg_ms"], "inv", k["k_col_inv"]["avg_ms"], j["integer_lag_exact"], j["other_workload"]["value"])
PY
done
unset TWX_LIB
