#!/bin/bash
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --latency-steps 0 --no-kernel-table --steps 400 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-56s %10.0f  %.4f ms' % ('$label', r['value'], r['ms_per_step']))"
}
for i in 1 2; do
for pad in 0 8192 16384 24576 40960; do
run "fulldec mel pad $pad" LYRA_HIP_LDS_PAD_logmel_noise=$pad -- --full-decoder
done
for pad in 0 8192 16384 32768; do
run "48k resample pad $pad" LYRA_HIP_LDS_PAD_resample=$pad -- --rate 48000
done
done
