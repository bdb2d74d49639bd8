mkdir -p gpurun_out/r5c
timeout -k 10 600 python -m pytest tests/test_gather_gpu.py -x -q -s > gpurun_out/r5c/gather.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c/gather.log; grep -E "^gather|passed|failed|Error|error|assert" gpurun_out/r5c/gather.log | tail -45
BENCH_ARGS="--variant plain" tools/micro/variants_ab2.sh r5c 1000000 2 k0=build_variants/lib_k0.so k1=build_variants/lib_k1.so k2=build_variants/lib_k2.so k3=build_variants/lib_k3.so
