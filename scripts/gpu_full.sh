#!/bin/bash
# the round-end checks in one call: full GPU suite, smoke, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -40 gpurun_out/full_tests.log; exit 1; }
tail -2 gpurun_out/full_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err || { tail gpurun_out/full_bench.err; exit 1; }
python -c "
import json; j=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1]); print('bench', round(j['ms_per_step'],2), round(j['value'],1), j['roofline']['frac'], j['cpu_baseline']['value'])"
