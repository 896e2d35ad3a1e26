#!/bin/bash
# rocprofv3 passes of one bench.py command line: kernel trace + stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes), SQ cycle split, L2 hit/miss, vector L1 tag accesses against its read requests to L2.  usage: scripts/prof.sh <tag> <bench.py args...>
# Results: gpurun_out/prof_<tag>/{stats,fetch,write,sq,tcc}/p_results.db  →  tools/prof_summary.py, tools/sq_summary.py
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# one stream: with the render's two streams kernels overlap and a launch's start-to-end time includes sharing the machine
# (closest-hit launches read 7.0 ms instead of 6.4); the roofline prices the kernel alone (bench.py)
export PT_OVERLAP=0
run() {  # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 400 rocprofv3 "$@" -d $out/$name -o p -- python3 bench.py --no-cpu-baseline "${BARGS[@]}" > $out/$name.json 2> $out/$name.err || { echo "pass $name failed"; tail -5 $out/$name.err; exit 1; }
    echo "pass $name ok"
}
BARGS=("$@")
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
find $out -name '*.db' | head; du -sh $out
