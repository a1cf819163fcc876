#!/bin/bash
# The CPU oracle's own suite under AddressSanitizer + UndefinedBehaviorSanitizer (Java int wraparound is restated
# with unsigned arithmetic: a signed overflow anywhere in oracle/ would be a parity bug of the checker itself).
# Builds an instrumented oracle/libgpx_oracle.so, runs the oracle-side tests against it, restores the normal build.
# libstdc++ is preloaded next to libasan: the oracle throws (BufferUnderflow of the wire decoder), and ASan's
# __cxa_throw interceptor has to find the real one at start-up, before ctypes dlopens the library.
set -e
cd "$(dirname "$0")/.."
trap 'rm -f oracle/libgpx_oracle.so; make -C oracle -s' EXIT
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared \
    -o oracle/libgpx_oracle.so oracle/gpx_oracle.cpp
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle_kat.py tests/test_wire_oracle.py tests/test_election_oracle.py \
    tests/test_host_rows_oracle.py tests/test_host_cluster_oracle.py -q -x -p no:cacheprovider "$@"
