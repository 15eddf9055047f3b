#!/bin/bash
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 400 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %10.0f  %.4f ms' % ('$label', r['value'], r['ms_per_step']))"
}
for i in 1 2 3; do
for m in "--rate 48000" "--rate 48000 --dtx" "--rate 48000 --full-decoder" "--rate 48000 --full-decoder --dtx" "--rate 8000"; do
run "ahead on sq: $m" X=1 -- $m
run "on se:       $m" LYRA_HIP_RS_IN_ON_SE=1 -- $m
done; done
