#!/bin/bash
# pair-kernel occupancy variants (launch bounds 4 / 6 / 8 waves per SIMD) on the soup and on C2
set -e
mkdir -p gpurun_out
L=gpurun_out/r2i.log; : > $L
for w in base pw6 pw8; do
  lib=build_ab/lib_$w.so; [ $w = base ] && lib=pytracer_amd/csrc/libpytracer_hip.so
  for wl in c4 c2; do
    echo "== $w $wl" >> $L
    PYTRACER_HIP_LIB=$lib PT_OVERLAP=0 timeout -k 10 200 python scripts/ab_bench.py --workload $wl --rounds 2 \
       --env PT_PAIR_LDS_LEVELS=12 --env PT_PAIR_LDS_LEVELS=8 >> $L 2>&1
  done
done
tail -50 $L
