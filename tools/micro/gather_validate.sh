# after a change to tgnh_gather.hip: the GPU suite, then the random-configuration soaks (plain, sharded, checkpoint), 3 minutes each
set -e
mkdir -p gpurun_out/g2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/g2/gpu_suite.log 2>&1
tail -2 gpurun_out/g2/gpu_suite.log
timeout -k 10 300 python tests/oracle_soak.py --minutes 3 --seed0 800000 > gpurun_out/g2/soak_plain.txt 2>&1
tail -4 gpurun_out/g2/soak_plain.txt
timeout -k 10 300 python tests/oracle_soak.py --sharded --minutes 3 --seed0 810000 > gpurun_out/g2/soak_sharded.txt 2>&1
tail -4 gpurun_out/g2/soak_sharded.txt
timeout -k 10 200 python tests/oracle_soak.py --checkpoint --minutes 2 --seed0 820000 > gpurun_out/g2/soak_ckpt.txt 2>&1
tail -4 gpurun_out/g2/soak_ckpt.txt
