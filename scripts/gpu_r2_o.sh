#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_c3_full.py tests/test_gpu_dropin.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2o_tests.log 2>&1 || { tail -30 gpurun_out/r2o_tests.log; exit 1; }
tail -2 gpurun_out/r2o_tests.log
PT_OVERLAP=0 timeout -k 10 200 python scripts/ab_bench.py --workload head --rounds 2 --env PT_TRACE_DYNAMIC=1 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2o_c3.json
python -c "
import json; j=json.loads(open('gpurun_out/r2o_c3.json').read().strip().splitlines()[-1]); print('bench ms', round(j['ms_per_step'],2), j['value'])"
