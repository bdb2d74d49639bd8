mkdir -p gpurun_out/r5d
timeout -k 10 600 python -m pytest tests/test_gather_gpu.py -q -s > gpurun_out/r5d/gather.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5d/gather.log; grep -E "passed|failed|Error|rc=" gpurun_out/r5d/gather.log | tail -5
BENCH_ARGS="--variant plain" tools/micro/variants_ab2.sh r5d 1000000 2 k0=build_variants/lib_k0.so d3=build_variants/lib_d3.so d4=build_variants/lib_d4.so
timeout -k 10 600 python3 tools/constrained_table.py r05 > gpurun_out/r5d/constrained.log 2>&1; echo "constrained rc=$?"; tail -5 gpurun_out/r5d/constrained.log
