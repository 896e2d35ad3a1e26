#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "dynamic" > gpurun_out/r2t_tests.log 2>&1 || { tail -30 gpurun_out/r2t_tests.log; exit 1; }
tail -2 gpurun_out/r2t_tests.log
PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload head --rounds 2 --env PT_TRACE_PACKET=0 --env PT_TRACE_PACKET=1 --env PT_TRACE_PACKET=2 --env PT_TRACE_PACKET=3 2>&1 | grep -v amdgpu.ids
