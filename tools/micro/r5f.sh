mkdir -p gpurun_out/r5f
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5f/gputests.log; tail -4 gpurun_out/r5f/gputests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 --variant resident > gpurun_out/r5f/gaps_resident.log 2>&1; echo "gaps rc=$?"
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 125000 --variant resident --dist > gpurun_out/r5f/gaps_rccl.log 2>&1; echo "gaps rccl rc=$?"
timeout -k 10 300 python3 tools/step_gaps.py r05 --molecules 1000000 --variant resident > gpurun_out/r5f/gaps_5M.log 2>&1; echo "gaps 5M rc=$?"
timeout -k 10 400 python tests/oracle_soak.py --minutes 4 --seed0 500000 > gpurun_out/r5f/oracle_soak.txt 2>&1; tail -12 gpurun_out/r5f/oracle_soak.txt
