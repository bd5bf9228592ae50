cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(j['value'], j['roofline']['frac'], j['roofline']['avg_ms'], j['cpu_baseline']['value'], j['integer_lag_exact'])"
