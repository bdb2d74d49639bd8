#!/bin/bash
# (How round 4's pattern-mass-constants experiment was run; the kernels it compares exist with profiles/r04_pconst_experiment.patch
# applied -- the experiment was taken out of the library again, profiles/r04_scaling_ceiling.md.)
# SQ counters of wstep_kernel at the 8-GPU shard size with the pattern mass constants (the library as built) and with per-slot
# reciprocals (build_variants/nopconst.so, -DTGNH_NO_PCONST).   usage: tools/micro/pconst_counters.sh <outdir>
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$1; mkdir -p $out
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
for v in table rcp; do
  lib=""; [ $v = rcp ] && lib=$root/build_variants/nopconst.so
  TGNH_LIB=$lib rocprofv3 --pmc $C --kernel-trace -d $out/$v --output-format csv -- python3 $root/bench.py --molecules 125000 --variant resident --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $out/$v.json 2> $out/$v.err || exit 1
done
python3 - $out <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for v in ("table", "rcp"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "wstep_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x) / len(x), 1) for k, x in sorted(acc.items())}, "launches", len(next(iter(acc.values()))) if acc else 0)
PY
