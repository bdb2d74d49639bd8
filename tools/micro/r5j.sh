mkdir -p gpurun_out/r5j
timeout -k 10 400 python tests/oracle_soak.py --minutes 5 --seed0 700000 --sharded > gpurun_out/r5j/oracle_soak--sharded.txt 2>&1; tail -4 gpurun_out/r5j/oracle_soak--sharded.txt | cut -c1-400; grep -c "^FAIL" gpurun_out/r5j/oracle_soak--sharded.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5j/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5j/gputests.log; tail -4 gpurun_out/r5j/gputests.log
