# run on the GPU box: per-kernel times of library variants (built by tools/variants.sh)
for v in "" $@; do
  lib=${v:+amaranth_twstft_amd/variants/lib_$v.so}
  echo -n "${v:-default}: "; TWX_LIB=$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --windows 64 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print({x:k[x]['avg_ms'] for x in k}, d['value'], d['integer_lag_exact'])"
done
