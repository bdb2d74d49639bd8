#!/bin/bash
# variants of the one-launch step at the 8-GPU shard size, interleaved on one box (each a tools/build_variant.py library):
# usage: [BENCH_ARGS="--variant plain"] tools/micro/variants_ab2.sh <outdir> <molecules> <rounds> name=lib[:ENV=VAL] ...
out=gpurun_out/$1; mol=$2; rounds=$3; shift 3
mkdir -p $out
for i in $(seq 1 $rounds); do for v in "$@"; do
name=${v%%=*}; rest=${v#*=}; lib=${rest%%:*}; envs=""
if [[ "$rest" == *:* ]]; then envs=${rest#*:}; fi
env TGNH_LIB=$lib $envs timeout -k 10 200 python bench.py --molecules $mol ${BENCH_ARGS:---variant resident --graph on --steps 3000 --warmup 100} --no-extra --no-cpu-baseline 2>>$out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mol $name', d['roofline']['kernel'], d['value'], d['integrator_only']['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $out/ab.txt
done; done
