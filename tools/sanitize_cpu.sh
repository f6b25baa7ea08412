#!/bin/bash
# CPU sanitizer job ("race detection / sanitizers"): the host build of the device headers (tests/host_emul/cg_emul.cpp:
# every LDS layout and index computation of the kernels, 1-thread workgroup shim) and the C oracle compiled with
# -fsanitize=address,undefined, and the CPU tests that drive them -- all six golden sizes up to the n = 57 layouts -- run against
# those builds.  GPU sanitizers are not available on this pool.   usage: tools/sanitize_cpu.sh [LOGFILE]
set -e
cd "$(dirname "$0")/.."
LOG=${1:-profiles/cpu_sanitizers.log}
python -m coulombgas_amd.build --sanitize --force
export CG_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{ echo "# $(date -u +%FT%TZ)  g++/gcc $(gcc -dumpversion)  -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined"
  echo "# libraries: tests/host_emul/libcg_emul_asan.so, oracle/_build/libcg_oracle_asan.so (CG_SANITIZE=1 selects them in coulombgas_amd/build.py)"
  python -m pytest tests/test_host_emul.py tests/test_oracle_kat.py tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider 2>&1
} | tee "$LOG"
