for cfg in "8 2" "4 2" "4 3" "4 4" "8 3" "2 4"; do
  set -- $cfg
  echo -n "batch=$1 streams=$2: "; TWX_STREAMS=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --windows 96 --batch $1 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['integer_lag_exact'])"
done
