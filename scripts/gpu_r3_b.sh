#!/bin/bash
# round 3, second GPU call: the new full-size parity tests, per-launch times of shards, kernel trace of shard renders
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_c3_full.py tests/test_c4_soup.py -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r3b_tests.log 2>&1 || { tail -60 gpurun_out/r3b_tests.log; exit 1; }
tail -3 gpurun_out/r3b_tests.log
timeout -k 10 300 python scripts/shard_launches.py 1 4 8 > gpurun_out/r3b_shard_launches.txt 2> gpurun_out/r3b.err || { tail gpurun_out/r3b.err; exit 1; }
cat gpurun_out/r3b_shard_launches.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3b_trace -o p -- python3 bench.py --steps 6 --warmup 2 --shard-of 8 --no-cpu-baseline > gpurun_out/r3b_trace.json 2> gpurun_out/r3b_trace.err || { tail gpurun_out/r3b_trace.err; exit 1; }
python3 tools/gap_summary.py r03b_shard8 gpurun_out/r3b_trace/p_results.db > gpurun_out/r3b_gaps.txt 2>&1 || tail gpurun_out/r3b_gaps.txt
cat gpurun_out/r3b_gaps.txt
rm -rf gpurun_out/r3b_trace
echo done
