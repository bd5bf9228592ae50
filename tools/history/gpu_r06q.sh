# round 6, GPU call Q: k_rowd_bandsum — the wave reduce-scatter checked alone, the carrier-search tests, tasks per thread 1 / 2 / 4 against k_rowd<BAND>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; rm -rf $O; mkdir -p $O
timeout 60 tools/bin/permlane_probe 2>&1 | tee $O/permlane.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for v in off "" bs1 bs4 off "" bs1 bs4; do
  lib=""; bs=1
  [ "$v" = off ] && bs=0
  [ -n "$v" ] && [ "$v" != off ] && lib=amaranth_twstft_amd/variants/lib_$v.so
  TWX_BANDSUM=$bs TWX_LIB=$lib timeout 120 python3 tools/kernel_alone.py k_row_band 3 2>&1 | tail -1 | sed "s/^/alone variant=$v /" | tee -a $O/ab.txt
done
for v in 0 1 0 1; do
  TWX_BANDSUM=$v timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caf --no-wideband --no-pmc > $O/bench_$v.json 2>$O/bench_$v.err
  python3 - "$v" <<'PY' | tee -a $O/ab.txt
import json, sys
j = json.loads([l for l in open("gpurun_out/r06q/bench_%s.json" % sys.argv[1]) if l.startswith("{")][-1])
print("bench bandsum=%s" % sys.argv[1], j["value"], j["other_workload"]["value"], j["integer_lag_exact"], j.get("kernels", {}).get("k_row_band"), j["roofline"]["avg_ms"])
PY
done
