for v in default r10_w4 r10_w8; do
  if [ $v = default ]; then unset TWX_LIB; else export TWX_LIB=$PWD/amaranth_twstft_amd/libtwx_$v.so; fi
  echo -n "VARIANT $v: "; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --windows 32 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_row_mid'], d['kernels']['k_row_band'], d['value'], d['integer_lag_exact'])"
done
