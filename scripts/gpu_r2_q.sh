#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_c4_soup.py tests/test_c2_head_loop.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r2q_tests.log 2>&1 || { tail -30 gpurun_out/r2q_tests.log; exit 1; }
tail -2 gpurun_out/r2q_tests.log
for wl in c4 c2; do
PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload $wl --rounds 2 --env PT_PAIR_REFILL=0 --env PT_PAIR_REFILL=1 2>&1 | grep -v amdgpu.ids
done
