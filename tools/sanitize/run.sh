#!/bin/bash
# ThreadSanitizer and AddressSanitizer + UBSan runs of the host SBVH builder (CPU build only: GPU sanitizers are not available on the pool).
set -e
cd "$(dirname "$0")"
SRC="builder_main.cpp ../../gmu-path-tracer_amd/host/sbvh_builder.cpp"
export GMUPT_BUILD_THREADS=8 GMUPT_BUILD_FANOUT=512
g++ -std=c++17 -O1 -g -fsanitize=thread -ffp-contract=off -o /tmp/gmupt_builder_tsan $SRC -lpthread
TSAN_OPTIONS="halt_on_error=1" /tmp/gmupt_builder_tsan 20000
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -o /tmp/gmupt_builder_asan $SRC -lpthread
ASAN_OPTIONS="detect_leaks=1" /tmp/gmupt_builder_asan 20000
echo "sanitizers: clean"
