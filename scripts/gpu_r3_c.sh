#!/bin/bash
# lobe-set shade kernels: schedule / parity tests, then A/B (one stream, per-stage HIP events) with the diffuse kernel
# built for 3 and for 2 waves per SIMD
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_schedules.py tests/test_gpu_parity.py tests/test_c4mini.py tests/test_specmini.py tests/test_texmini.py tests/test_xtra.py tests/test_c4_soup.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3c_tests.log 2>&1 || { tail -60 gpurun_out/r3c_tests.log; exit 1; }
tail -3 gpurun_out/r3c_tests.log
for lib in libpytracer_hip.so libpt_w2.so; do
  echo "== $lib"
  PYTRACER_HIP_LIB=pytracer_amd/csrc/$lib timeout -k 10 300 python scripts/ab_bench.py --env PT_NO_HIT_CLASSES=1,PT_OVERLAP=0 --env PT_OVERLAP=0 --rounds 3 2>&1 | grep -v amdgpu.ids || exit 1
  PYTRACER_HIP_LIB=pytracer_amd/csrc/$lib timeout -k 10 300 python scripts/ab_bench.py --workload c4 --env PT_NO_HIT_CLASSES=1,PT_OVERLAP=0 --env PT_OVERLAP=0 --rounds 2 2>&1 | grep -v amdgpu.ids || exit 1
  PYTRACER_HIP_LIB=pytracer_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c_$lib.json 2> gpurun_out/r3c.err || { tail gpurun_out/r3c.err; exit 1; }
  python -c "
import json; j=json.loads(open('gpurun_out/r3c_$lib.json').read().strip().splitlines()[-1]); print('bench', round(j['ms_per_step'],2), round(j['value'],1), j['roofline']['exclusive']['stage_ms_per_step'])"
done
echo done
