for v in 1 0; do
  echo -n "TWX_ROWD=$v: "; TWX_STREAMS=1 TWX_ROWD=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --windows 64 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print({x:k[x]['avg_ms'] for x in k}, d['value'], d['integer_lag_exact'])"
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
