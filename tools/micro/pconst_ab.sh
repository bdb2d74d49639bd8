#!/bin/bash
# (How round 4's pattern-mass-constants experiment was run; the kernels it compares exist with profiles/r04_pconst_experiment.patch
# applied -- the experiment was taken out of the library again, profiles/r04_scaling_ceiling.md.)
# Pattern mass constants (PCONST_*, tgnh_internal.h) against the per-slot reciprocals (a -DTGNH_NO_PCONST build,
# tools/build_variant.py build_variants/nopconst.so -DTGNH_NO_PCONST), interleaved on one box: the 8-GPU shard (625 k slots,
# hipGraph) and the metric size (eager, bench.py's default shape).   usage: tools/micro/pconst_ab.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
one() {  # label lib molecules extra-args...
  local label=$1 lib=$2 m=$3; shift 3
  TGNH_LIB=$lib timeout -k 10 200 python bench.py --molecules $m --variant resident --no-extra --no-cpu-baseline "$@" 2>> $out/ab.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', $m, d['config']['variant_ran'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $out/ab.txt
}
for i in 1 2 3; do
  one table "" 125000 --graph on --steps 3000 --warmup 200 || exit 1
  one rcp $PWD/build_variants/nopconst.so 125000 --graph on --steps 3000 --warmup 200 || exit 1
done
for i in 1 2; do
  one table "" 1000000 --steps 1000 --warmup 100 || exit 1
  one rcp $PWD/build_variants/nopconst.so 1000000 --steps 1000 --warmup 100 || exit 1
done
