#!/bin/bash
# ray binning (k_sort_rays) on the 10^6-triangle soup: parity with the sort forced on, then A/B of keys and windows
set -o pipefail
mkdir -p gpurun_out
PT_SORT_RAYS=3 PT_TRACE_PAIR=3 timeout -k 10 600 python -u -m pytest tests/test_c4_soup.py tests/test_gpu_schedules.py tests/test_c4mini.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e_tests.log 2>&1 || { tail -40 gpurun_out/r3e_tests.log; exit 1; }
tail -2 gpurun_out/r3e_tests.log
timeout -k 10 600 python scripts/ab_bench.py --workload c4 --rounds 2 --env PT_OVERLAP=0 --env PT_OVERLAP=0,PT_SORT_RAYS=1 --env PT_OVERLAP=0,PT_SORT_RAYS=2 --env PT_OVERLAP=0,PT_SORT_RAYS=3 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_MODE=1 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_MODE=2 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_MODE=80 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_WINDOW=16384 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_WINDOW=16384,PT_SORT_MODE=80 --env PT_OVERLAP=0,PT_SORT_RAYS=3,PT_SORT_WINDOW=1024 2>&1 | grep -v amdgpu.ids
echo done
