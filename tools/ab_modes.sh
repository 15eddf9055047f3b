#!/bin/bash
# GPU box: the bench's optional legs under environment settings, alternating:  bash tools/ab_modes.sh
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 400 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %10.0f  %.4f ms' % ('$label', r['value'], r['ms_per_step']))"
}
for i in 1 2; do
run "all3 grouped-sn (new)" X=1 -- --full-decoder --dtx --rate 48000
run "all3 split-sn (old)" LYRA_HIP_SPLIT_SN_CALLS=1 -- --full-decoder --dtx --rate 48000
run "fulldec+48k grouped" X=1 -- --full-decoder --rate 48000
run "fulldec+48k split" LYRA_HIP_SPLIT_SN_CALLS=1 -- --full-decoder --rate 48000
run "48k prio default" X=1 -- --rate 48000
run "48k prio e0,d1,q2" LYRA_HIP_PRIO=0,1,2 -- --rate 48000
run "48k prio e0,d2,q2" LYRA_HIP_PRIO=0,2,2 -- --rate 48000
run "48k prio e1,d0,q2" LYRA_HIP_PRIO=1,0,2 -- --rate 48000
run "48k prio flat" LYRA_HIP_FLAT_PRIO=1 -- --rate 48000
run "bare" X=1 --
done
