#!/bin/bash
# CPU only: the oracle (the checker every parity claim rests on) built with AddressSanitizer + UndefinedBehaviorSanitizer, the CPU
# test files that drive it run against that build, the normal build restored afterwards.  (GPU sanitizers are not available on the
# pool; the HIP library's host code is exercised by the same tests on the GPU box without them.)
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -std=c11 -fPIC -fopenmp -ffp-contract=off -fno-fast-math $(grep -qw fma /proc/cpuinfo && echo -mfma) \
    -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o oracle/_build/libenvbuild_oracle.so oracle/envbuild_oracle.c -lm
trap 'make -s -B -C oracle' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_oracle_golden.py tests/test_oracle_env_step.py tests/test_policy_oracle.py tests/test_callers_host.py \
    tests/test_abi_and_host.py tests/test_sharding_gloo.py tests/test_bench_contract.py -q -m "not gpu" -s 2>&1 | tee /tmp/eb_sanitize.log | tail -3
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/eb_sanitize.log)"
