#!/bin/bash
# The host build of the stepper (oracle/arena_port.cpp over csrc/arena_*.h) under AddressSanitizer + UBSan, once with the host's own layouts
# and once with the DEVICE's LDS layouts (-DRLG_TICKWORK_OVERLAY -DRLG_QUEUE_LEAVES: car tick context over the contact list, candidate
# leaves instead of slots), through every CPU test of tests/test_oracle_golden.py (tapes, one-tick pairs, gym rollouts, setters, meshes).
# GPU sanitizers are not available on this pool; this is where an out-of-bounds index of the shared source would show.   tools/asan_port.sh
set -e
cd "$(dirname "$0")/.."
ASAN_LIB=$(gcc -print-file-name=libasan.so); UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
cp oracle/_build/liboracle_port.so /tmp/liboracle_port_backup.so
trap 'cp /tmp/liboracle_port_backup.so oracle/_build/liboracle_port.so' EXIT
for defs in "" "-DRLG_TICKWORK_OVERLAY=1 -DRLG_QUEUE_LEAVES=1"; do
  echo "== host build with: ${defs:-the host layouts}"
  g++ -std=c++17 -O1 -g -fPIC -ffp-contract=off $defs -fsanitize=address,undefined -fno-omit-frame-pointer -shared oracle/arena_port.cpp rlgymppo_cpp_amd/csrc/arena_mesh.cpp -o oracle/_build/liboracle_port.so -lm -lpthread
  LD_PRELOAD="$ASAN_LIB $UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 python -m pytest tests/test_oracle_golden.py -q -m "not gpu" 2>&1 | grep -E "runtime error|ERROR: AddressSanitizer|passed|failed|SUMMARY" | sort | uniq -c
done
