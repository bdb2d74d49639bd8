#!/bin/bash
# Builds tgnh_host.cpp with AddressSanitizer + UBSan (g++, CPU only) and runs the host-logic tests and the malformed-descriptor /
# odd-argument fuzz of the boundary against it.
set -e
cd "$(dirname "$0")/../.."
OUT=/tmp/libdrudetgnh_hostasan.so
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include openmm_drudenose_amd/csrc/tgnh_host.cpp tools/sanitize/launch_stubs.cpp \
    -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib -o $OUT
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN:$UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    TGNH_LIB=$OUT python -m pytest tests/test_host_logic.py tests/test_desc_fuzz.py -q -x -p no:cacheprovider "$@"
