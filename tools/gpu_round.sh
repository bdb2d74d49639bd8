#!/bin/bash
# One GPU call's worth of checks: the GPU suite, then (only if pytest itself ran to its end: rc 0 or 1) the bench lines.
# usage: tools/gpu_round.sh <outdir> [pytest args...]
out=gpurun_out/$1; shift
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > $out/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> $out/gputests.log
tail -4 $out/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $out/bench_default.json 2> $out/bench_default.err || exit 1
python - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default:", d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for k, v in d.get("extra", {}).items():
    print("  ", k, v["steps_per_s"], {n: r["avg_us"] for n, r in v["kernels"].items()})
PY
for m in 125000 250000; do
timeout -k 10 200 python bench.py --molecules $m --variant resident --graph on --steps 2000 --warmup 100 --no-extra --no-cpu-baseline > $out/shard_$m.json 2> $out/shard_$m.err || exit 1
python -c "import json,sys; d=json.loads(open('$out/shard_$m.json').read().strip().splitlines()[-1]); print('$m', d['config']['variant_ran'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
