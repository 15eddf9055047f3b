#!/bin/bash
# GPU box: A/B of library builds on ONE box for the full decoder step (decode + decoder-side NoiseEstimator), alternating:
#   tools/ab_full_decoder.sh <variant.so> [rounds]
V=$1; R=${2:-3}
for i in $(seq $R); do for lib in "" $V; do LYRA_HIP_LIB=$lib python bench.py --full-decoder --no-cpu-baseline --steps 400 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default}', r['value'], r['ms_per_step'], 'logmel_noise', r['kernels'].get('logmel_noise_kernel',{}).get('avg_us'))"; done; done
