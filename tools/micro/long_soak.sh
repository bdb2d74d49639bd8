# a longer soak of the final binary (one GPU call, ~17 minutes: gpurun allows 20): random configurations against the oracle (one handle, sharded,
# checkpoint) and random call sequences with half of the walks on the gather path
set -e
base=${1:-1000000}     # seed base: another one gives another sample
out=gpurun_out/long_soak; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -k "full_size_steps" > $out/full_size.log 2>&1; tail -2 $out/full_size.log
timeout -k 10 700 python tests/oracle_soak.py --minutes 6 --seed0 $base > $out/soak_plain.txt 2>&1; tail -4 $out/soak_plain.txt | cut -c1-400
timeout -k 10 500 python tests/oracle_soak.py --sharded --minutes 3 --seed0 $((base + 100000)) > $out/soak_sharded.txt 2>&1; tail -1 $out/soak_sharded.txt | cut -c1-400
timeout -k 10 300 python tests/oracle_soak.py --checkpoint --minutes 1.5 --seed0 $((base + 200000)) > $out/soak_ckpt.txt 2>&1; tail -1 $out/soak_ckpt.txt | cut -c1-400
timeout -k 10 600 python tools/fuzz_soak.py --gather --modes --constrained --minutes 4 --seed0 $((base + 300000)) > $out/fuzz_gather.txt 2>&1; tail -1 $out/fuzz_gather.txt
