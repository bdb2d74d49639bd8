mkdir -p gpurun_out/r5l
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5l/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5l/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5l/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5l/gputests.log; tail -3 gpurun_out/r5l/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 tools/profile_round.py r05 > gpurun_out/r5l/profile_round.log 2>&1; echo "profile rc=$?"
timeout -k 10 600 python3 tools/profile_round.py r05_shard625k --molecules 125000 --skip-sq > gpurun_out/r5l/profile_shard.log 2>&1; echo "profile shard rc=$?"
for v in "--variant resident" "--variant resident --dist" "--variant defer"; do timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 $v > /dev/null 2>&1; echo "gaps $v rc=$?"; done
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 1000000 --variant resident > /dev/null 2>&1; echo "gaps 5M rc=$?"
TGNH_LIB=build_variants/lib_trace.so timeout -k 10 120 python tools/step_trace.py 125000 1000000 > gpurun_out/r5l/step_trace_mixed.txt 2>&1; echo "trace rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5l/bench_driver_shape.json 2> gpurun_out/r5l/bench_driver_shape.err; echo "bench driver shape rc=$?"
timeout -k 10 300 python bench.py --no-extra > gpurun_out/r5l/bench_default_run.json 2> gpurun_out/r5l/bench_default_run.err; echo "bench default rc=$?"
for m in 1000000 125000; do timeout -k 10 300 python bench.py --molecules $m --variant plain-gather --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m plain-gather', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r5l/gather_vs_tiled.txt
timeout -k 10 300 python bench.py --molecules $m --variant plain --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m plain', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r5l/gather_vs_tiled.txt; done
python -c "
import json
for f in ('bench_driver_shape','bench_default_run'):
    d=json.loads(open('gpurun_out/r5l/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['integrator_only']['value'], d['roofline']['device_copy_GBps'], d.get('cpu_baseline',{}).get('value'))"
