for v in 0 -1; do
  echo -n "ROW_PERSISTENT=$v: "; TWX_ROW_PERSISTENT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --windows 64 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_row_mid'], d['value'], d['integer_lag_exact'])"
done
TWX_ROW_PERSISTENT=-1 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
