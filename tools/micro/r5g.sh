mkdir -p gpurun_out/r5g
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5g/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5g/gputests.log; tail -4 gpurun_out/r5g/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
(TGNH_LIB=build_variants/lib_k0.so timeout -k 10 300 python tools/micro/dualnh_quirk.py; timeout -k 10 300 python tools/micro/dualnh_quirk.py) > gpurun_out/r5g/dualnh_quirk.txt 2>&1; grep dualNH gpurun_out/r5g/dualnh_quirk.txt
bash tools/scaling_ceiling.sh > gpurun_out/r5g/scaling_ceiling.txt 2>&1; tail -14 gpurun_out/r5g/scaling_ceiling.txt
