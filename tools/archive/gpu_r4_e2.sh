#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
run() { timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 600 --latency-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))"; }
for i in 1 2; do
for b in 184 120 64 8; do run --streams 4096 --bits $b; done
done | tee gpurun_out/r04/bits_sweep.txt
