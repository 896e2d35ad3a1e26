#!/bin/bash
# the whole GPU suite + smoke (what the round end runs), then the default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/gpu_tests_bench.json 2> gpurun_out/gpu_tests_bench.err || { tail gpurun_out/gpu_tests_bench.err; exit 1; }
python -c "
import json; j=json.loads(open('gpurun_out/gpu_tests_bench.json').read().strip().splitlines()[-1]); print('bench', round(j['ms_per_step'],2), round(j['value'],1), round(j['roofline']['frac'],3), round(j['roofline']['exclusive']['frac'],3), j['cpu_baseline']['value'])"
