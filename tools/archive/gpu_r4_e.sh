#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
run() { timeout 300 python bench.py --no-cpu-baseline --no-kernel-table --steps 600 --latency-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SKIP_GUARDS=$LYRA_HIP_UNSAFE_SKIP_GUARDS $*', r['value'], r['ms_per_step'], r['roofline']['frac'], 'verified', r.get('verified'))"; }
for i in 1 2 3; do
for f in 0 1; do export LYRA_HIP_UNSAFE_SKIP_GUARDS=$f
run --config 3
done
done | tee gpurun_out/r04/guard_wait_cost.txt
