#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r2g_tests.log 2>&1 || { tail -30 gpurun_out/r2g_tests.log; exit 1; }
tail -2 gpurun_out/r2g_tests.log
timeout -k 10 300 python scripts/ab_bench.py --env PT_OVERLAP=0 --env PT_OVERLAP=1 --rounds 2 > gpurun_out/r2g_ab_head.log 2>&1 || { tail -20 gpurun_out/r2g_ab_head.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2g_ab_head.log
for o in 0 1; do PT_OVERLAP=$o timeout -k 10 300 python bench.py --steps 8 --warmup 2 --shard-of 2,4,8 --no-cpu-baseline > gpurun_out/r2g_proxy_$o.json 2>gpurun_out/r2g_proxy.err || { tail gpurun_out/r2g_proxy.err; exit 1; }; done
for w in c2 c4; do for o in 0 1; do PT_OVERLAP=$o timeout -k 10 300 python bench.py --workload $w --spp 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_${w}_$o.json 2>gpurun_out/r2g_w.err || { tail gpurun_out/r2g_w.err; exit 1; }; done; done
echo done
