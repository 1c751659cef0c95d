#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (CPU build only; the GPU pool offers no sanitizer runs):
# builds oracle/sgpr_oracle.c with -fsanitize=address,undefined into a temporary liborc.so, runs the golden
# tests of the oracle against it, and puts the regular build back.
set -e
cd "$(dirname "$0")/.."
make -C oracle liborc.so > /dev/null
cp oracle/liborc.so /tmp/liborc_regular.so
gcc -O1 -g -fopenmp -fPIC -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o oracle/liborc.so oracle/sgpr_oracle.c -lm
trap 'cp /tmp/liborc_regular.so oracle/liborc.so' EXIT
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py -x -q
