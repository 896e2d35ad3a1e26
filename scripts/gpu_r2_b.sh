#!/bin/bash
# pair-kernel parity (variant test + axis rays under PT_TRACE_PAIR=3) then A/B on head, c2, c4
set -o pipefail
mkdir -p gpurun_out
PT_TRACE_PAIR=3 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_c4mini.py tests/test_specmini.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2b_tests.log 2>&1 || { tail -40 gpurun_out/r2b_tests.log; exit 1; }
tail -3 gpurun_out/r2b_tests.log
timeout -k 10 300 python scripts/ab_bench.py --env PT_TRACE_PAIR=0 --env PT_TRACE_PAIR=3 --env PT_TRACE_PAIR=1 --env PT_TRACE_PAIR=2 --rounds 2 > gpurun_out/r2b_ab_head.log 2>&1 || { tail -20 gpurun_out/r2b_ab_head.log; exit 1; }
cat gpurun_out/r2b_ab_head.log
timeout -k 10 300 python scripts/ab_bench.py --workload c2 --env PT_TRACE_PAIR=0 --env PT_TRACE_PAIR=3 --rounds 2 > gpurun_out/r2b_ab_c2.log 2>&1 || { tail -20 gpurun_out/r2b_ab_c2.log; exit 1; }
cat gpurun_out/r2b_ab_c2.log
timeout -k 10 400 python scripts/ab_bench.py --workload c4 --env PT_TRACE_PAIR=0 --env PT_TRACE_PAIR=3 --rounds 2 > gpurun_out/r2b_ab_c4.log 2>&1 || { tail -20 gpurun_out/r2b_ab_c4.log; exit 1; }
cat gpurun_out/r2b_ab_c4.log
