#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --workload c4 --spp 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3d_c4.json 2> gpurun_out/r3d.err || { tail gpurun_out/r3d.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r3d_c4.json').read().strip().splitlines()[-1])
print('c4 16spp', round(j['ms_per_step'], 2), round(j['value'], 1))
for e in j['roofline']['per_launch']:
    print(e['symbol'], e['level'], e['rays'], round(e['node_visits_per_ray'], 1), round(e['tri_tests_per_ray'], 2), round(e['ms'], 3), round(e['Grays_per_s'], 3), round(e['frac'], 3))
print(j['roofline']['exclusive']['stage_ms_per_step'])
PY
for e in PT_TRACE_META=1 PT_TRACE_META=2 PT_TRACE_META=1 PT_TRACE_META=2; do env $e timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$e\", round(j[\"ms_per_step\"],2), j['roofline']['exclusive']['stage_ms_per_step'])" || exit 1; done
echo done
