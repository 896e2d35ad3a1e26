#!/bin/bash
set -e
for lib in pytracer_amd/csrc/libpytracer_hip.so build_ab/lib_pk6.so build_ab/lib_pk8.so; do
echo "== $lib"
PYTRACER_HIP_LIB=$lib PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload head --rounds 2 --env PT_TRACE_PACKET=3 2>&1 | grep round
done
