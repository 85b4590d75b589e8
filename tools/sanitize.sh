#!/bin/bash
# tools/sanitize.sh [OUTDIR]: the CPU builds under sanitizers (the pool has no GPU sanitizer; this container has no GPU at all).
#   1. the oracle (oracle/*.hpp, oracle_capi.cpp) with AddressSanitizer + UBSan + libstdc++'s container assertions, then every CPU test
#      that uses it against that build (tests/conftest.py: ORACLE_LIB); libasan is preloaded into python, leak checking off (the
#      interpreter's own allocations would drown the report);
#   2. the library's host-side logic - what tests/test_host_logic_sanitized.py builds and runs in the CPU suite.
# Takes about four minutes.  The GPU tests can use an assertions-only build of the oracle (no sanitizer runtime in the process):
#   ORACLE_LIB=$OUT/liboracle_assert.so python -m pytest tests -m gpu
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/oracle/_san}
mkdir -p "$OUT"
FLAGS="-march=x86-64-v3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -D_GLIBCXX_ASSERTIONS"
g++ -O1 -g $FLAGS -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared -o "$OUT/liboracle_san.so" "$R/oracle/oracle_capi.cpp"
g++ -O2 $FLAGS -shared -o "$OUT/liboracle_assert.so" "$R/oracle/oracle_capi.cpp"
cd "$R"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 ORACLE_LIB="$OUT/liboracle_san.so" \
    python -m pytest tests/test_equil.py tests/test_oracle_assembly.py tests/test_oracle_cpr.py tests/test_oracle_ecl_output.py tests/test_oracle_endscale.py \
    tests/test_oracle_hysteresis.py tests/test_oracle_linalg.py tests/test_oracle_pvt.py -q -p no:cacheprovider
python -m pytest tests/test_host_logic_sanitized.py -q -p no:cacheprovider
