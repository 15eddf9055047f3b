#!/bin/bash
# GPU box: decoder chain one priority level up (LYRA_HIP_PRIO=0,1,2) per leg, alternating; bare step 3 regions each to see bimodality
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 400 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-56s %10.0f  %.4f ms' % ('$label', r['value'], r['ms_per_step']))"
}
for i in 1 2 3 4; do
for m in "" "--full-decoder" "--rate 48000" "--dtx" "--rate 48000 --full-decoder --dtx" "--config 4" "--config 5"; do
run "prio 0,0,2 (default): $m" X=1 -- $m
run "prio 0,1,2          : $m" LYRA_HIP_PRIO=0,1,2 -- $m
done; done
