#!/bin/bash
# chunk cursors (PT_TRACE_DYNAMIC) against the static stride: parity tests, then the shard proxy both ways
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "dynamic or variants" > gpurun_out/r2j_tests.log 2>&1 || { tail -30 gpurun_out/r2j_tests.log; exit 1; }
tail -3 gpurun_out/r2j_tests.log
for d in 0 1; do
  PT_TRACE_DYNAMIC=$d timeout -k 10 200 python bench.py --steps 10 --warmup 2 --shard-of 2,4,8 --no-cpu-baseline > gpurun_out/r2j_proxy_$d.json 2> gpurun_out/r2j_proxy_$d.err
  PT_TRACE_DYNAMIC=$d PT_OVERLAP=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --shard-of 8 --no-cpu-baseline > gpurun_out/r2j_proxy_noovl_$d.json 2> gpurun_out/r2j_proxy_noovl_$d.err
done
python - <<'PY'
import json
for f in ('proxy_0','proxy_1','proxy_noovl_0','proxy_noovl_1'):
    j=json.loads(open('gpurun_out/r2j_%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(j['ms_per_step'],2), {k:(round(v['ms_per_step'],2), round(v['predicted_efficiency'],3)) for k,v in j['shard_proxy'].items() if isinstance(v,dict)})
PY
