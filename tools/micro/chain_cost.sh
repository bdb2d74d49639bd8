# what a chain of 3 links costs per step by size: inside the streaming launches (TGNH_INLINE_MULTI_MAX large) or as chain_kernel (0)
mkdir -p gpurun_out/r3g
export TGNH_LIB=$PWD/build_variants/tuning.so
for m in 6400 25000 125000 400000 1000000; do for v in defer resident; do for mm in 0 100000000; do
TGNH_INLINE_MULTI_MAX=$mm python bench.py --molecules $m --chains 3 --variant $v --graph on --steps 300 --warmup 30 --no-extra --no-cpu-baseline > gpurun_out/r3g/cc_${m}_${v}_${mm}.json 2>gpurun_out/r3g/cc_${m}_${v}_${mm}.err || exit 1
python - <<PY
import json
b=json.loads(open("gpurun_out/r3g/cc_${m}_${v}_${mm}.json").read().strip().splitlines()[-1])
print($m,"$v","inline" if $mm else "chain_kernel",b["value"],b["ms_per_step"], b["config"]["variant_ran"], {k:(v["avg_us"],v["launches"]) for k,v in b["kernels"].items()})
PY
done; done; done
