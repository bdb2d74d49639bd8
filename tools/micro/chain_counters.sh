# chain_kernel (3 links): instructions per launch over a run -- does the careful path (the polynomial's range left) run?
mkdir -p gpurun_out/r3g && export TMPDIR=/tmp
export TGNH_LIB=$PWD/build_variants/tuning.so TGNH_INLINE_MULTI_MAX=0
for m in 125000 1000000; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/r3g/cc_pmc_$m --output-format csv -- python3 bench.py --molecules $m --chains 3 --variant defer --graph off --steps 30 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r3g/cc_pmc_$m.out 2> gpurun_out/r3g/cc_pmc_$m.err || { tail -5 gpurun_out/r3g/cc_pmc_$m.err; exit 1; }
python3 - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/r3g/cc_pmc_$m/**/*counter_collection.csv", recursive=True):
    rows+=[r for r in csv.DictReader(open(f)) if "chain_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="SQ_INSTS_VALU"]
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
print([int(float(r["Counter_Value"])) for r in rows])
PY
done
