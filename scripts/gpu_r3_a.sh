#!/bin/bash
# round 3, first GPU call: the whole GPU suite on the new tree, the headline bench with per-launch census,
# the shard proxy and C1 without events, and a kernel trace of shard renders for the gap analysis
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_zz_bench_contract.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3a_tests.log 2>&1 || { tail -60 gpurun_out/r3a_tests.log; exit 1; }
tail -3 gpurun_out/r3a_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r3a_c3.json 2> gpurun_out/r3a.err || { tail gpurun_out/r3a.err; exit 1; }
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --shard-of 2,4,8 --no-cpu-baseline > gpurun_out/r3a_proxy.json 2>> gpurun_out/r3a.err || { tail gpurun_out/r3a.err; exit 1; }
timeout -k 10 300 python bench.py --res 128 --spp 1 --integrator direct --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3a_c1.json 2>> gpurun_out/r3a.err || { tail gpurun_out/r3a.err; exit 1; }
# kernel trace of shard-0-of-8 renders (one stream and two): where the time between kernels goes
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3a_trace -o p -- python3 bench.py --steps 6 --warmup 2 --shard-of 8 --no-cpu-baseline > gpurun_out/r3a_trace.json 2> gpurun_out/r3a_trace.err || { tail gpurun_out/r3a_trace.err; exit 1; }
python3 tools/gap_summary.py r03a_shard8 gpurun_out/r3a_trace/p_results.db > gpurun_out/r3a_gaps.txt 2>&1 || tail gpurun_out/r3a_gaps.txt
cp profiles/r03a_shard8_gaps.csv gpurun_out/ 2>/dev/null
python3 - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/r3a_trace/p_results.db')
print([r[1] for r in con.execute("pragma table_info('kernels')")])
PY
rm -rf gpurun_out/r3a_trace
python - <<'PY'
import json
for f in ('r3a_c3', 'r3a_proxy', 'r3a_c1'):
    j = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
    print(f, round(j['ms_per_step'], 3), round(j['value'], 1), j['roofline'].get('frac'), (j['roofline'].get('exclusive') or {}).get('frac'))
    if 'shard_proxy' in j:
        print({k: (round(v['ms_per_step'], 3), round(v['predicted_efficiency'], 3)) for k, v in j['shard_proxy'].items() if isinstance(v, dict)})
    if f == 'r3a_c3':
        print(json.dumps(j['roofline'].get('per_symbol'), indent=1))
PY
echo done
