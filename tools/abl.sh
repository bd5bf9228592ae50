for v in default nokeep; do
  if [ $v = default ]; then unset TWX_LIB; else export TWX_LIB=$PWD/amaranth_twstft_amd/libtwx_$v.so; fi
  for i in 1 2; do echo -n "VARIANT $v: "; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --windows 64 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_row_mid'], d['value'], d['integer_lag_exact'])"; done
done
