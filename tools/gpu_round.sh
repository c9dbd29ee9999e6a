#!/bin/bash
# One GPU visit: parity tests, smoke, bench, kernel-trace profile. Steps are chained with && and
# each has its own time limit; output goes to gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py --steps 8 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
