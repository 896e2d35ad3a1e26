#!/bin/bash
# round-2 first GPU pass: whole GPU suite, the WIP no-alloc test, headline bench, shard proxy, 2-rank gloo rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r2a_tests.log 2>&1 || { tail -30 gpurun_out/r2a_tests.log; exit 1; }
tail -3 gpurun_out/r2a_tests.log
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2a_noalloc.log 2>&1 || { tail -30 gpurun_out/r2a_noalloc.log; exit 1; }
tail -3 gpurun_out/r2a_noalloc.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err || { tail -20 gpurun_out/r2a_bench.err; exit 1; }
cat gpurun_out/r2a_bench.json
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --shard-of 2,4,8 --no-cpu-baseline > gpurun_out/r2a_proxy.json 2> gpurun_out/r2a_proxy.err || { tail -20 gpurun_out/r2a_proxy.err; exit 1; }
cat gpurun_out/r2a_proxy.json
PT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2a_gloo2.json 2> gpurun_out/r2a_gloo2.err || { tail -20 gpurun_out/r2a_gloo2.err; exit 1; }
cat gpurun_out/r2a_gloo2.json
