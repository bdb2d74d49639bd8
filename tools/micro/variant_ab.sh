# defer against resident at the metric size, interleaved on one box (bench.py default shape: eager, 1000 steps)
mkdir -p gpurun_out/r3j
for i in 1 2 3; do for v in defer resident; do
python bench.py --variant $v --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done; done
