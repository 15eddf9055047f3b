#!/bin/bash
# GPU box: alternate an environment setting on the shipped library, driver form + sustained:
#   tools/ab_env.sh <rounds> <regions> VAR=a VAR=b ...
R=$1; N=$2; shift 2
for i in $(seq $R); do for kv in "$@"; do
  echo -n "$kv  "; env $kv timeout 200 python tools/k20_repeat.py $N $BENCH_ARGS 2>&1 | tail -1
done; done
