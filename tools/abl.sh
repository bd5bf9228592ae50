for v in default f1 f2 f3; do
  if [ $v = default ]; then unset TWX_LIB; else export TWX_LIB=$PWD/amaranth_twstft_amd/libtwx_$v.so; fi
  echo -n "VARIANT $v: "; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --windows 64 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print({x:k[x]['avg_ms'] for x in k}, d['value'], d['integer_lag_exact'])"
done
