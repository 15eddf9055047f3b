#!/bin/bash
# GPU box: driver form (K = 20) + sustained form for several library builds on ONE box, alternating:
#   tools/ab_k20.sh <rounds> <regions per process> [bench args --] lib1.so lib2.so ...   ("default" = the shipped library)
R=$1; N=$2; shift 2
for i in $(seq $R); do for lib in "$@"; do
  L=$lib; [ "$lib" = default ] && L=""
  LYRA_HIP_LIB=$L timeout 200 python tools/k20_repeat.py $N $BENCH_ARGS 2>&1 | tail -1
done; done
