#!/bin/bash
# rocprofv3 passes for the headline, C2 and C4 (scripts/prof.sh), summarised ON the box into profiles/ (tools/*_summary.py);
# only the summaries travel back (gpurun_out/profiles_out/): the raw databases exceed what gpurun merges.
# usage: scripts/prof_all.sh <round tag, e.g. r02>
set -e
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag, bench args
    local t=$1; shift
    bash scripts/prof.sh $t "$@"
    local d=gpurun_out/prof_$t extra=""
    [ $t = head ] && extra=headline
    python3 tools/prof_summary.py ${R}_$t $d/stats/p_results.db $d/fetch/p_results.db $d/write/p_results.db $extra
    python3 tools/sq_summary.py ${R}_$t $d/sq/p_results.db
    python3 tools/tcc_summary.py ${R}_$t $d/tcc/p_results.db
    mkdir -p gpurun_out/profiles_out/json_$t
    cp $d/*.json gpurun_out/profiles_out/json_$t/
    rm -rf $d
}
prof head --steps 3 --warmup 1
prof c2 --workload c2 --steps 5 --warmup 2
prof c4 --workload c4 --spp 16 --steps 2 --warmup 1
cp profiles/${R}_head_* profiles/${R}_c2_* profiles/${R}_c4_kernel* profiles/${R}_c4_pmc* profiles/${R}_c4_sq* profiles/${R}_c4_tcc* profiles/pmc_summary.json gpurun_out/profiles_out/
ls gpurun_out/profiles_out
