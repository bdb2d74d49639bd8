mkdir -p gpurun_out/r5e
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5e/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5e/gputests.log; tail -4 gpurun_out/r5e/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 --variant resident > gpurun_out/r5e/gaps_resident.log 2>&1; echo "gaps rc=$?"
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 --variant resident --dist > gpurun_out/r5e/gaps_rccl.log 2>&1; echo "gaps rccl rc=$?"
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 --variant defer > gpurun_out/r5e/gaps_defer.log 2>&1; echo "gaps defer rc=$?"
tail -12 gpurun_out/r5e/gaps_resident.log
