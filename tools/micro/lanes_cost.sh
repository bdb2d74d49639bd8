# chains of 10 links: a link per lane (chain_lanes_run) or LDS-resident links, by size
mkdir -p gpurun_out/r3h
export TGNH_LIB=$PWD/build_variants/tuning.so
for m in 6400 125000 1000000; do for l in 0 1; do
TGNH_CHAIN_LANES=$l python bench.py --molecules $m --chains 10 --variant defer --graph on --steps 300 --warmup 30 --no-extra --no-cpu-baseline > gpurun_out/r3h/lanes_${m}_${l}.json 2>gpurun_out/r3h/lanes_${m}_${l}.err || exit 1
python - <<PY
import json
b=json.loads(open("gpurun_out/r3h/lanes_${m}_${l}.json").read().strip().splitlines()[-1])
print($m,"lanes" if $l else "LDS links",b["value"],b["ms_per_step"], {k:(v["avg_us"],v["launches"]) for k,v in b["kernels"].items()})
PY
done; done
