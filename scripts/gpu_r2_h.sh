#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_shade_f32.py -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/r2h_tests.log 2>&1 || true
grep "f32 shade,\|passed\|failed" gpurun_out/r2h_tests.log
PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --env F32=0 --env F32=1 --rounds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2h_ab.log
timeout -k 10 300 python scripts/ab_bench.py --env F32=0 --env F32=1 --rounds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2h_ab_ovl.log
