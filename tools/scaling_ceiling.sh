for m in 1000000 500000 250000 125000; do
  for v in defer resident; do
    python bench.py --molecules $m --variant $v --graph on --steps 2000 --warmup 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', '$v', d['config']['variant_ran'], '-', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  done
  TGNH_FORCE_DIST=1 python bench.py --molecules $m --variant resident --graph on --steps 2000 --warmup 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'resident', d['config']['variant_ran'], 'rccl', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
