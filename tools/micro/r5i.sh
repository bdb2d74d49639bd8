mkdir -p gpurun_out/r5i
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5i/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5i/gputests.log; tail -4 gpurun_out/r5i/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
(TGNH_LIB=build_variants/lib_k0.so timeout -k 10 300 python tools/micro/dualnh_quirk.py; timeout -k 10 300 python tools/micro/dualnh_quirk.py) > gpurun_out/r5i/dualnh_quirk.txt 2>&1; grep "10 links" gpurun_out/r5i/dualnh_quirk.txt
for mode in "" "--sharded" "--checkpoint"; do
timeout -k 10 300 python tests/oracle_soak.py --minutes 3 --seed0 700000 $mode > gpurun_out/r5i/oracle_soak$mode.txt 2>&1; tail -7 gpurun_out/r5i/oracle_soak$mode.txt | cut -c1-400
done
