mkdir -p gpurun_out/r5k
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5k/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r5k/gputests.log; tail -4 gpurun_out/r5k/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 tools/profile_round.py r05 > gpurun_out/r5k/profile_round.log 2>&1; echo "profile rc=$?"; tail -25 gpurun_out/r5k/profile_round.log
timeout -k 10 600 python3 tools/profile_round.py r05_shard625k --molecules 125000 --skip-sq > gpurun_out/r5k/profile_shard.log 2>&1; echo "profile shard rc=$?"
