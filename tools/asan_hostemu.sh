#!/bin/bash
# The host-side C++ the product shares with the test emulation (sf_tables.cpp: table builder; sf_stages.h /
# sf_math.h: every stage) under AddressSanitizer + UBSan, driven by the hostemu parity tests.
cd "$(dirname "$0")/.."
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" HOSTEMU_SANITIZE=1 \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_hostemu.py -x -q "$@"
