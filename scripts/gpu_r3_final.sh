#!/bin/bash
# round-3 bench lines: all five BASELINE.json configurations, the scaling proxy, the 2-rank gloo rehearsal (N > 1 control flow)
set -o pipefail
mkdir -p gpurun_out
run() { local out=$1; shift; timeout -k 10 400 "$@" > gpurun_out/$out 2>> gpurun_out/r3z.err || { tail gpurun_out/r3z.err; exit 1; }; }
: > gpurun_out/r3z.err
run r3z_c3.json python bench.py --steps 20 --warmup 5
run r3z_c1.json python bench.py --res 128 --spp 1 --integrator direct --steps 200 --warmup 20 --no-cpu-baseline
run r3z_c2.json python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline
run r3z_c4.json python bench.py --workload c4 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline
run r3z_c5.json python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline
run r3z_proxy.json python bench.py --steps 8 --warmup 2 --shard-of 2,4,8 --no-cpu-baseline
PT_BENCH_BACKEND=gloo run r3z_gloo2.json python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 3 --warmup 1
python - <<'PY'
import json
for f in ('c3', 'c1', 'c2', 'c4', 'c5', 'proxy', 'gloo2'):
    j = json.loads(open('gpurun_out/r3z_%s.json' % f).read().strip().splitlines()[-1])
    r = j['roofline']
    print(f, 'ms', round(j['ms_per_step'], 3), 'Mrays/s', round(j['value'], 1), 'Msamples/s', round(j['samples_per_s'] / 1e6, 1), 'frac', r.get('frac'), 'excl', (r.get('exclusive') or {}).get('frac'))
    if 'shard_proxy' in j:
        print({k: (round(v['ms_per_step'], 3), round(v['predicted_efficiency'], 3)) for k, v in j['shard_proxy'].items() if isinstance(v, dict)})
    if 'collective' in j:
        print(j['collective'], j['config'].get('checksum_rel_diff'))
    if f in ('c2', 'c4', 'c5'):
        for k, v in (r.get('per_symbol') or {}).items():
            print('   ', k, v['launches'], round(v['Grays_per_s'], 3), round(v['frac'], 3))
PY
