#!/bin/bash
# as variants_ab.sh, at the metric size (5 M slots, eager, 1 000 steps):  tools/micro/variants_ab_metric.sh <outdir> name=path.so ...
out=gpurun_out/$1; shift; mkdir -p $out
for i in 1 2 3; do
  for v in "$@"; do
    name=${v%%=*}; lib=${v#*=}; [ -n "$lib" ] && lib=$PWD/$lib
    TGNH_LIB=$lib timeout -k 10 200 python bench.py --variant resident --no-extra --no-cpu-baseline --steps 1000 --warmup 100 2>> $out/abm.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['config']['variant_ran'], d['value'], d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $out/abm.txt || exit 1
  done
done
