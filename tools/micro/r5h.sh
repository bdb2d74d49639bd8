mkdir -p gpurun_out/r5h
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5h/gputests.log; tail -4 gpurun_out/r5h/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
# defer against resident at the metric size, interleaved on one box: the driver's shape (eager, 20 after 5) and steady state (eager, 1000 after 100)
for i in 1 2 3; do for v in defer resident; do for shape in "--steps 20 --warmup 5" "--steps 1000 --warmup 100"; do
python bench.py --variant $v --no-extra --no-cpu-baseline $shape 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$shape', d['value'], d['integrator_only']['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/r5h/variant_ab.txt
done; done; done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5h/bench_driver_shape.json 2> gpurun_out/r5h/bench_driver_shape.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r5h/bench_driver_shape.json').read().strip().splitlines()[-1]); print(d['value'], d['integrator_only'], d['config']['harness_force'], list(d['extra'].keys())); print({k:v['steps_per_s'] for k,v in d['extra'].items()})"
