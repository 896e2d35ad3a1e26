#!/bin/bash
# hybrid-stack pair kernel: parity under several LDS depths (2 = nearly everything spills), then A/B on c4 / c2 / head
set -o pipefail
mkdir -p gpurun_out
for L in 2 12; do
PT_PAIR_LDS_LEVELS=$L PT_TRACE_PAIR=3 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_c4mini.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2c_tests_$L.log 2>&1 || { tail -40 gpurun_out/r2c_tests_$L.log; exit 1; }
tail -2 gpurun_out/r2c_tests_$L.log
done
timeout -k 10 400 python scripts/ab_bench.py --workload c4 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=64 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=16 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=12 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=8 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=4 --rounds 2 > gpurun_out/r2c_ab_c4.log 2>&1 || { tail -20 gpurun_out/r2c_ab_c4.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2c_ab_c4.log
timeout -k 10 300 python scripts/ab_bench.py --workload c2 --env PT_TRACE_PAIR=0 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=64 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=12 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=8 --rounds 2 > gpurun_out/r2c_ab_c2.log 2>&1 || { tail -20 gpurun_out/r2c_ab_c2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2c_ab_c2.log
timeout -k 10 300 python scripts/ab_bench.py --env PT_TRACE_PAIR=0 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=64 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=12 --env PT_TRACE_PAIR=3,PT_PAIR_LDS_LEVELS=8 --rounds 2 > gpurun_out/r2c_ab_head.log 2>&1 || { tail -20 gpurun_out/r2c_ab_head.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2c_ab_head.log
