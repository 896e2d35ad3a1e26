#!/bin/bash
set -e
for m in 4 8 16 24 32 48; do
echo "== PT_REFILL_MIN=$m"
PT_REFILL_MIN=$m PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload c4 --rounds 1 --env PT_PAIR_REFILL=1 2>&1 | grep "round"
PT_REFILL_MIN=$m PT_OVERLAP=0 timeout -k 10 300 python scripts/ab_bench.py --workload c2 --rounds 1 --env PT_PAIR_REFILL=1 2>&1 | grep "round"
done
