#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r2s_tests.log 2>&1 || { tail -30 gpurun_out/r2s_tests.log; exit 1; }
tail -2 gpurun_out/r2s_tests.log
timeout -k 10 400 python bench.py --workload c4 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2s_c4.json
timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2s_c2.json
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2s_c3.json
python - <<'PY'
import json
for t in ('c4','c2','c3'):
    j=json.loads(open('gpurun_out/r2s_%s.json'%t).read().strip().splitlines()[-1]); r=j['roofline']
    print(t, round(j['ms_per_step'],2), round(j['value'],1), round(r['frac'],3), r['kernel'], {k:round(v,2) for k,v in r['stage_ms_per_step'].items()})
PY
