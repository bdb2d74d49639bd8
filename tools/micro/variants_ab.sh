#!/bin/bash
# Several builds of the library interleaved on one box, the one-launch step at the 8-GPU shard size (625 k slots, hipGraph):
#   tools/micro/variants_ab.sh <outdir> name=path.so [name=path.so ...]      ("name=" alone: the library as built)
out=gpurun_out/$1; shift; mkdir -p $out
for i in 1 2 3; do
  for v in "$@"; do
    name=${v%%=*}; lib=${v#*=}; [ -n "$lib" ] && lib=$PWD/$lib
    TGNH_LIB=$lib timeout -k 10 200 python bench.py --molecules 125000 --variant resident --no-extra --no-cpu-baseline --graph on --steps 3000 --warmup 200 2>> $out/ab.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['config']['variant_ran'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $out/ab.txt || exit 1
  done
done
