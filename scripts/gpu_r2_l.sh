#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/r2l.log; : > $L
for ov in 0 1; do
echo "== head overlap $ov" >> $L
PT_OVERLAP=$ov timeout -k 10 300 python scripts/ab_bench.py --workload head --rounds 3 --env PT_TRACE_DYNAMIC=1,PT_SHADE_DYNAMIC=0 --env PT_TRACE_DYNAMIC=1,PT_SHADE_DYNAMIC=1 >> $L 2>&1
done
echo "== c5" >> $L
timeout -k 10 300 python scripts/ab_bench.py --workload c5 --rounds 2 --env PT_TRACE_DYNAMIC=1,PT_SHADE_DYNAMIC=0 --env PT_TRACE_DYNAMIC=1,PT_SHADE_DYNAMIC=1 >> $L 2>&1
grep -v amdgpu.ids $L
