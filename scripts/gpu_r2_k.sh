#!/bin/bash
# chunk cursors on the large-tree workloads (pair kernel: claim when needed) and on C5
set -e
mkdir -p gpurun_out
L=gpurun_out/r2k.log; : > $L
for lib in pytracer_amd/csrc/libpytracer_hip.so build_ab/lib_pw5.so; do
  for wl in c4 c2; do
    echo "== $lib $wl" >> $L
    PYTRACER_HIP_LIB=$lib PT_OVERLAP=0 timeout -k 10 200 python scripts/ab_bench.py --workload $wl --rounds 2 \
       --env PT_TRACE_DYNAMIC=0 --env PT_TRACE_DYNAMIC=1 >> $L 2>&1
  done
done
echo "== c5" >> $L
timeout -k 10 300 python scripts/ab_bench.py --workload c5 --rounds 2 --env PT_TRACE_DYNAMIC=0 --env PT_TRACE_DYNAMIC=1 >> $L 2>&1
grep -v amdgpu.ids $L
