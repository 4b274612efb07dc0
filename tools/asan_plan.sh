#!/bin/bash
# The device-free host code of the library (symbolic analysis, replay tables, plan export: csrc/jg_symbolic.cpp + jg_plan_api.cpp) under
# AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool, the CPU build is):
#   tools/asan_plan.sh [pytest args]        builds build/libjg_plan_asan.so and runs tests/test_plan_cpu.py against it
# The CPU suite runs the same thing through tests/test_sanitizer_cpu.py.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -o build/libjg_plan_asan.so juliagrid.jl_amd/csrc/jg_symbolic.cpp juliagrid.jl_amd/csrc/jg_plan_api.cpp
ASAN=$(g++ -print-file-name=libasan.so)
UBSAN=$(g++ -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 JG_PLAN_LIB=$PWD/build/libjg_plan_asan.so \
    python -m pytest tests/test_plan_cpu.py -x -q -p no:cacheprovider "$@"
