set -e
mkdir -p gpurun_out/g1
timeout -k 10 500 python -m pytest tests/test_gather_gpu.py -x -q > gpurun_out/g1/tests.log 2>&1
tail -3 gpurun_out/g1/tests.log
for m in 1000000 125000; do for v in plain-gather plain; do
timeout -k 10 200 python bench.py --molecules $m --variant $v --steps 200 --no-extra --no-cpu-baseline > gpurun_out/g1/b_${m}_$v.json 2>gpurun_out/g1/b_${m}_$v.err || true
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/g1/b_${m}_$v.json").read().strip().splitlines()[-1])
    print($m,"$v",round(d["value"],1),{k:v["avg_us"] for k,v in d["kernels"].items()})
except Exception as e: print("fail",e)
PY
done; done
