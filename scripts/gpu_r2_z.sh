#!/bin/bash
# round-2 final numbers: all five BASELINE.json configurations, the scaling proxy, the 2-rank gloo rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2z_c3.json 2> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 300 python bench.py --res 128 --spp 1 --integrator direct --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2z_c1.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2z_c2.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 400 python bench.py --workload c4 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2z_c4.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 300 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2z_c5.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --shard-of 2,4,8 --no-cpu-baseline > gpurun_out/r2z_proxy.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
PT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2z_gloo2.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --shade-f32 --no-cpu-baseline > gpurun_out/r2z_c3_f32.json 2>> gpurun_out/r2z.err || { tail gpurun_out/r2z.err; exit 1; }
echo done
