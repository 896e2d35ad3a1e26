#!/bin/bash
# LDS-resident tree kernel: parity, then A/B on the head mesh (headline workload)
set -o pipefail
mkdir -p gpurun_out
PT_TRACE_LDS=3 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_c4mini.py tests/test_specmini.py tests/test_xtra.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2d_tests.log 2>&1 || { tail -40 gpurun_out/r2d_tests.log; exit 1; }
tail -2 gpurun_out/r2d_tests.log
timeout -k 10 300 python scripts/ab_bench.py --env PT_TRACE_LDS=0 --env PT_TRACE_LDS=3 --env PT_TRACE_LDS=1 --env PT_TRACE_LDS=2 --rounds 2 > gpurun_out/r2d_ab_head.log 2>&1 || { tail -20 gpurun_out/r2d_ab_head.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2d_ab_head.log
