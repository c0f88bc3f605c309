#!/bin/bash
# scripts/san_cpu.sh -- the HOST side under AddressSanitizer + UndefinedBehaviorSanitizer, on a CPU box:
#   * libbn_mi355x_san.so (make -C bayesiannetwork_amd/csrc SAN=1): planners (bn_plan.cpp, bn_small_plan.cpp, bn_dag_plan.cpp), engine
#     (bn_engine*.cpp) and the sampler's host driver (bn_lw.cpp), exercised by tests/test_host_logic.py on BN_DEVICE_HOST_ONLY engines;
#   * liboracle_san.so (make -C oracle SAN=1): the checker, exercised by tests/test_oracle_golden.py;
#   * the header-only drop-in (flatten, DSC loader) in tests/cpp/test_dropin.cpp --flatten / --dsc ... --flatten, over include/compat and,
#     where /root/reference exists, over the reference's own graph.hpp / matrix.hpp / serializer.
# No GPU is touched and none is needed (GPU sanitizers do not exist on this pool).  Log: the first argument, default profiles/san_cpu.log.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/san_cpu.log}
cd "$ROOT"
{
echo "== san_cpu.sh  $(date -u +%Y-%m-%dT%H:%M:%SZ)  g++ $(g++ -dumpversion)  HEAD $(git rev-parse --short HEAD 2>/dev/null)"
make -C bayesiannetwork_amd/csrc -j4 all > /dev/null && make -C bayesiannetwork_amd/csrc SAN=1 -j4 2>&1 | tail -1
make -C oracle SAN=1 2>&1 | grep -v "^make" | tail -1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
ASAN_RT=$(gcc -print-file-name=libasan.so)
rc=0
echo "== tests/test_host_logic.py + tests/test_oracle_golden.py with the sanitizer builds of both libraries"
LD_PRELOAD=$ASAN_RT BN_MI355X_LIB=$ROOT/bayesiannetwork_amd/libbn_mi355x_san.so BN_ORACLE_LIB=$ROOT/oracle/liboracle_san.so \
    python -m pytest tests/test_host_logic.py tests/test_oracle_golden.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4 || rc=1
[ "${PIPESTATUS[0]}" = 0 ] || rc=1
mkdir -p build/san
for model in include/compat /root/reference; do
    [ -d "$model" ] || { echo "== $model absent: skipped"; continue; }
    exe=build/san/dropin_$(basename "$model")
    echo "== tests/cpp/test_dropin.cpp over $model: --flatten, --dsc tests/golden/alarm_shaped.dsc --flatten"
    g++ -std=c++14 -O1 -g -Wall -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I include -I "$model" \
        tests/cpp/test_dropin.cpp -L bayesiannetwork_amd -lbn_mi355x_san -Wl,-rpath,$ROOT/bayesiannetwork_amd -Wl,-rpath,/opt/rocm/lib -o "$exe" || { rc=1; continue; }
    "$exe" --flatten > /dev/null || { echo "FAILED: $exe --flatten"; rc=1; }
    "$exe" --dsc tests/golden/alarm_shaped.dsc --flatten > /dev/null || { echo "FAILED: $exe --dsc"; rc=1; }
    echo "   ok"
done
echo "== bench_dropin --checksum (graph_t / cpt_t construction of BASELINE configs[0..2], flatten, position table)"
g++ -std=c++14 -O1 -g -Wall -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I include -I include/compat \
    tests/cpp/bench_dropin.cpp -L bayesiannetwork_amd -lbn_mi355x_san -Wl,-rpath,$ROOT/bayesiannetwork_amd -Wl,-rpath,/opt/rocm/lib -o build/san/bench_dropin \
    && build/san/bench_dropin --checksum > /dev/null && echo "   ok" || { echo "FAILED: bench_dropin --checksum"; rc=1; }
echo "== result: $([ $rc = 0 ] && echo CLEAN || echo FAILED)"
} 2>&1 | tee "$LOG"
grep -q "== result: CLEAN" "$LOG"
