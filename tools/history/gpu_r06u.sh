# round 6, GPU call U: k_rowd<BAND, double> with resident workgroups (next row loaded ahead) — parity, then the fp64 leg against the one-row form
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "f64 or double or precision or wideband or sweep or two_forms or full_size" 2>&1 | tail -5 | tee $O/pytest.txt
for v in nopb "" nopb "" pf1 pf5; do
  lib=""; pf=2
  [ "$v" = nopb ] && lib=amaranth_twstft_amd/variants/lib_nopb.so
  [ "$v" = pf1 ] && pf=1
  [ "$v" = pf5 ] && pf=5
  TWX_BAND_PF=$pf TWX_LIB=$lib timeout 300 python3 bench.py --wideband-only > $O/wb_$v.json 2>/dev/null
  python3 - "$v" <<'PY' | tee -a $O/ab.txt
import json, sys
ls = [json.loads(l) for l in open("gpurun_out/r06u/wb_%s.json" % sys.argv[1]) if l.startswith("{")]
for j in ls:
    f = j.get("f64_workload") or (j if "correlated_Msamples_per_s" in j and j.get("dtype") == "f64" else None)
    if f: print("variant=%s" % sys.argv[1], f.get("correlated_Msamples_per_s"), f.get("roofline", {}).get("frac"), {k: v.get("avg_ms") for k, v in (f.get("kernels") or {}).items()})
PY
done
