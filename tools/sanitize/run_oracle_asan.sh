#!/bin/bash
# Builds the CPU oracle with AddressSanitizer + UBSan and runs the oracle tests against it (CPU only).
set -e
cd "$(dirname "$0")/../.."
cp oracle/libtgnh_oracle.so /tmp/libtgnh_oracle.keep 2>/dev/null || true
gcc -O1 -g -fPIC -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared \
    -o oracle/libtgnh_oracle.so oracle/tgnh_oracle.c oracle/water_ff.c -lm
trap 'make -s -B -C oracle libtgnh_oracle.so >/dev/null' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests/test_oracle.py tests/test_host_logic.py -q -x -p no:cacheprovider "$@"
