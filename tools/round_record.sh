#!/bin/bash
# The round's record on the FINAL binary, in one GPU call (~25 GPU-minutes): smoke, the GPU suite, then every profile that is
# stamped with the kernel-source hash (profile_round: rocprofv3 stats + PMC passes at the metric system and at an 8-GPU
# shard's size; the constrained step launch by launch; the profiler's view of the gaps between launches), the two bench
# shapes (the driver's, steady state) and the gather path beside the tiles.  Afterwards, where the repository is tracked:
#     cp gpurun_out/profiles_out/<tag>_* profiles/ ; cp gpurun_out/record/{bench_driver_shape,bench_default_run}.json profiles/<tag>_...
# No edit under csrc/ or to include/drude_tgnh.h after this without running it again: bench.py quotes roofline.traffic from
# <tag>_pmc_traffic.json only while its csrc_sha matches the library's.
#     usage: bash tools/round_record.sh r05
tag=${1:?tag, e.g. r05}
out=gpurun_out/record
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $out/smoke.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $out/gputests.log; tail -3 $out/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 tools/profile_round.py $tag > $out/profile_round.log 2>&1 || { echo "profile_round failed"; exit 1; }
timeout -k 10 600 python3 tools/profile_round.py ${tag}_shard625k --molecules 125000 --skip-sq > $out/profile_shard.log 2>&1 || { echo "profile_round (shard) failed"; exit 1; }
timeout -k 10 600 python3 tools/constrained_table.py $tag > $out/constrained.log 2>&1 || { echo "constrained_table failed"; exit 1; }
for v in "--variant resident" "--variant resident --dist" "--variant defer"; do
    timeout -k 10 300 python3 tools/step_gaps.py $tag --molecules 125000 $v > /dev/null 2>&1 || { echo "step_gaps $v failed"; exit 1; }
done
timeout -k 10 300 python3 tools/step_gaps.py $tag --molecules 1000000 --variant resident > /dev/null 2>&1 || { echo "step_gaps 5M failed"; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err || { echo "bench (driver shape) failed"; exit 1; }
timeout -k 10 300 python bench.py --no-extra > $out/bench_default_run.json 2> $out/bench_default_run.err || { echo "bench (default) failed"; exit 1; }
: > $out/gather_vs_tiled.txt
for m in 1000000 125000; do for v in plain-gather plain; do
    timeout -k 10 300 python bench.py --molecules $m --variant $v --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m $v', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" >> $out/gather_vs_tiled.txt || { echo "gather bench failed"; exit 1; }
done; done
cat $out/gather_vs_tiled.txt
python - <<PY
import json
for f in ('bench_driver_shape','bench_default_run'):
    d=json.loads(open('$out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['integrator_only']['value'], d.get('cpu_baseline',{}).get('value'), d.get('csrc_sha'))
PY
