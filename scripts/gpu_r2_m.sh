#!/bin/bash
# full GPU suite on the tree with chunk cursors on by default, then the headline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r2m_tests.log 2>&1 || { tail -40 gpurun_out/r2m_tests.log; exit 1; }
tail -3 gpurun_out/r2m_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2m_c3.json 2> gpurun_out/r2m.err || { tail gpurun_out/r2m.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r2m_c3.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['samples_per_s'], j['roofline']['frac'], j['roofline']['stage_ms_per_step'])
PY
