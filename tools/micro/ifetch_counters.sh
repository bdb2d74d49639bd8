# instruction-fetch counters of chain_kernel (3 links) at 32 k slots and at the metric size: does the serial chain wait for its own code?
mkdir -p gpurun_out/r3g && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INSTS_VALU\b" | sort -u > gpurun_out/r3g/ifetch_names.txt
cat gpurun_out/r3g/ifetch_names.txt | tr '\n' ' '; echo
for m in 6400 1000000; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/r3g/ifetch_$m --output-format csv -- python3 bench.py --molecules $m --chains 3 --variant defer --graph off --steps 40 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r3g/ifetch_$m.out 2> gpurun_out/r3g/ifetch_$m.err || { tail -5 gpurun_out/r3g/ifetch_$m.err; exit 1; }
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r3g/ifetch_$m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "chain_kernel" in n or "tile_kernel" in n or "wke_kernel" in n:
            acc[n.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    print($m, k, {c: round(sum(v)/len(v),1) for c,v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
done
