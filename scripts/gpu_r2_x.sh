#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_measmini.py tests/test_gpu_parity.py tests/test_texmini.py tests/test_gpu_dropin.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r2x_tests.log 2>&1 || { tail -30 gpurun_out/r2x_tests.log; exit 1; }
tail -2 gpurun_out/r2x_tests.log
for v in 0 1; do
if [ $v = 0 ]; then export PT_NO_HIT_CLASSES=1; else unset PT_NO_HIT_CLASSES; fi
timeout -k 10 300 python scripts/ab_bench.py --workload c5 --rounds 2 --env PT_OVERLAP=1 2>&1 | grep "round 2\|film"
done
