mkdir -p gpurun_out/r3g && export TMPDIR=/tmp
export TGNH_LIB=$PWD/build_variants/tuning.so TGNH_INLINE_MULTI_MAX=0
for m in 25000 125000; do
rocprofv3 --kernel-trace -d gpurun_out/r3g/cc_tr_$m --output-format csv -- python3 bench.py --molecules $m --chains 3 --variant defer --graph off --steps 100 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r3g/cc_tr_$m.out 2> gpurun_out/r3g/cc_tr_$m.err || { tail -5 gpurun_out/r3g/cc_tr_$m.err; exit 1; }
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/r3g/cc_tr_$m/**/*kernel_trace.csv", recursive=True):
    rows+=[r for r in csv.DictReader(open(f)) if "tgnh" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "chain_kernel" in r["Kernel_Name"]]
print($m, "chain_kernel n=%d durations us: min %.1f median %.1f max %.1f" % (len(d), min(d), sorted(d)[len(d)//2], max(d)))
for a,b in zip(rows[-9:-1], rows[-8:]):
    n=a["Kernel_Name"].split("(")[0][-32:]
    print(f"  {n:34s} dur {(int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3:8.2f} us   gap to next {(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3:8.2f} us")
PY
done
