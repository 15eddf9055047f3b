#!/bin/bash
# GPU box: background kernels (estimator, resamplers) confined to a subset of the CUs, alternating:  bash tools/ab_modes2.sh
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 400 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %10.0f  %.4f ms  %s' % ('$label', r['value'], r['ms_per_step'], r.get('verified')))"
}
for i in 1 2; do
run "fulldec default" X=1 -- --full-decoder
run "fulldec n=1/2" LYRA_HIP_CU_MASKS=0,0,0,00ff00ff -- --full-decoder
run "fulldec n=1/4" LYRA_HIP_CU_MASKS=0,0,0,000f000f -- --full-decoder
run "fulldec n=1/8" LYRA_HIP_CU_MASKS=0,0,0,00030003 -- --full-decoder
run "48k default" X=1 -- --rate 48000
run "48k n=1/4" LYRA_HIP_CU_MASKS=0,0,0,000f000f -- --rate 48000
run "48k q=1/4 n=1/4" LYRA_HIP_CU_MASKS=0,0,000f000f,000f000f -- --rate 48000
run "48k q=1/4(other) n=1/4" LYRA_HIP_CU_MASKS=0,0,00f000f0,000f000f -- --rate 48000
run "all3 default" X=1 -- --full-decoder --dtx --rate 48000
run "all3 n=1/4" LYRA_HIP_CU_MASKS=0,0,0,000f000f -- --full-decoder --dtx --rate 48000
run "all3 q=1/4(other) n=1/4" LYRA_HIP_CU_MASKS=0,0,00f000f0,000f000f -- --full-decoder --dtx --rate 48000
run "bare q=1/4" LYRA_HIP_CU_MASKS=0,0,00f000f0,0 -- 
run "bare" X=1 --
done
