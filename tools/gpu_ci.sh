#!/bin/bash
# GPU-box CI: parity tests, then (only if pytest did not time out) the micro-benchmarks.
mkdir -p gpurun_out
timeout -k 10 ${TEST_TIMEOUT:-700} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/tests.log
grep -E "passed|failed|error" gpurun_out/tests.log | tail -5
grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: skipping benches"; exit $rc; fi
if [ -n "${BENCH_ARGS:-}" ]; then
  timeout -k 10 300 python tools/bench_kernels.py ${BENCH_ARGS} > gpurun_out/bench_kernels.log 2>&1 && tail -60 gpurun_out/bench_kernels.log
fi
exit $rc
