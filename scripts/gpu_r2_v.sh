#!/bin/bash
set -e
for wl in c2 c4 c5; do
echo "== $wl"
PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload $wl --rounds 2 --env PT_TRACE_PACKET=0 --env PT_TRACE_PACKET=1 --env PT_TRACE_PACKET=3 2>&1 | grep "round 2"
done
