#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/r2n.log; : > $L
for lib in pytracer_amd/csrc/libpytracer_hip.so build_ab/lib_seg32.so build_ab/lib_seg128.so; do
  echo "== $lib" >> $L
  PYTRACER_HIP_LIB=$lib PT_OVERLAP=0 timeout -k 10 200 python scripts/ab_bench.py --workload head --rounds 2 --env PT_TRACE_DYNAMIC=1 >> $L 2>&1
  PYTRACER_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --shard-of 8 > gpurun_out/r2n_tmp.json 2>> $L
  python -c "
import json; j=json.loads(open('gpurun_out/r2n_tmp.json').read().strip().splitlines()[-1]); print('bench ms', round(j['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in j['shard_proxy'].items() if isinstance(v,dict)})" >> $L
done
grep -v amdgpu.ids $L
