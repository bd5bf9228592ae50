# round 6, GPU call O: batch size x pipeline slots on the round-6 kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; rm -rf $O; mkdir -p $O
for s in 3 2 4; do for b in 8 6 10 12 16; do
  TWX_STREAMS=$s timeout 200 python3 bench.py --steps 20 --warmup 3 --batch $b --no-cpu-baseline --no-caf --no-wideband --no-pmc --no-roofline > $O/b.json 2>/dev/null
  python3 - "$s" "$b" <<'PY' | tee -a $O/sweep.txt
import json, sys
try:
    j = json.loads([l for l in open("gpurun_out/r06o/b.json") if l.startswith("{")][-1])
    print("streams", sys.argv[1], "batch", sys.argv[2], j["value"], j["other_workload"]["value"], j["integer_lag_exact"])
except Exception as e:
    print("streams", sys.argv[1], "batch", sys.argv[2], "failed", e)
PY
done; done
