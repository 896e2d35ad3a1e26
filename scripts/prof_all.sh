#!/bin/bash
# rocprofv3 passes for the headline, C2, C4 (and C4 with ray binning on, C5 kernel stats only) (scripts/prof.sh), summarised
# ON the box into profiles/ (tools/*_summary.py); only the summaries travel back (gpurun_out/profiles_out/): the raw
# databases exceed what gpurun merges.   usage: scripts/prof_all.sh <round tag, e.g. r03> [workloads...]
set -e
R=${1:-r03}; shift || true
WL=${@:-head c2 c4 c4full c4sort c5 shard8 c1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/profiles_out
prof() {  # tag, bench args
    local t=$1; shift
    bash scripts/prof.sh $t "$@"
    local d=gpurun_out/prof_$t extra=""
    [ $t = head ] && extra=headline
    python3 tools/prof_summary.py ${R}_$t $d/stats/p_results.db $d/fetch/p_results.db $d/write/p_results.db $extra > /dev/null
    python3 tools/sq_summary.py ${R}_$t $d/sq/p_results.db > /dev/null
    python3 tools/tcc_summary.py ${R}_$t $d/tcc/p_results.db > /dev/null
    python3 tools/tcp_summary.py ${R}_$t $d/tcp/p_results.db > /dev/null
    mkdir -p gpurun_out/profiles_out/json_$t
    cp $d/*.json gpurun_out/profiles_out/json_$t/
    rm -rf $d
    cp profiles/${R}_${t}_* gpurun_out/profiles_out/
}
stats_only() {  # tag, env assignment or "", bench args: kernel trace + stats (+ gaps) only
    local t=$1 e=$2; shift; shift
    local d=gpurun_out/prof_$t
    mkdir -p $d
    ( [ -n "$e" ] && export $e; export PT_OVERLAP=${PT_OVERLAP_FOR_STATS:-0}; timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $d/stats -o p -- python3 bench.py --no-cpu-baseline "$@" > $d/stats.json 2> $d/stats.err ) || { echo "stats pass $t failed"; tail -5 $d/stats.err; exit 1; }
    python3 tools/prof_summary.py ${R}_$t $d/stats/p_results.db > /dev/null
    python3 tools/gap_summary.py ${R}_$t $d/stats/p_results.db > /dev/null
    rm -rf $d
    cp profiles/${R}_${t}_* gpurun_out/profiles_out/
}
for w in $WL; do
    case $w in
    head) prof head --steps 3 --warmup 1 ;;
    c2) prof c2 --workload c2 --steps 5 --warmup 2 ;;
    c4) prof c4 --workload c4 --spp 16 --steps 2 --warmup 1 ;;
    c4full) PT_PAIR_HALF=0 prof c4full --workload c4 --spp 16 --steps 2 --warmup 1 ;;   # full-size (128-byte) pair records
    c4sort) PT_SORT_RAYS=3 prof c4sort --workload c4 --spp 16 --steps 2 --warmup 1 ;;
    c5) stats_only c5 "" --workload c5 --steps 2 --warmup 1 ;;
    shard8) PT_OVERLAP_FOR_STATS=1 stats_only shard8 "" --steps 6 --warmup 2 --shard-of 8 ;;
    c1) PT_OVERLAP_FOR_STATS=1 stats_only c1 "" --res 128 --spp 1 --integrator direct --steps 30 --warmup 5 ;;
    esac
    echo "== $w done"
done
[ -f profiles/pmc_summary.json ] && cp profiles/pmc_summary.json gpurun_out/profiles_out/
ls gpurun_out/profiles_out
