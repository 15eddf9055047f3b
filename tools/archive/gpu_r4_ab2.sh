#!/bin/bash
# GPU box: full GPU suite, then A/B on ONE box: round-3 library (mode exact) vs this build (exact, xnnpack)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
run() { LYRA_HIP_LIB=$1 timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 400 --latency-steps 0 --requant $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${1:-default} requant=$2', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))"; }
for i in 1 2 3; do
  run lyra_amd/variants/r3.so exact
  run "" xnnpack
done | tee gpurun_out/r04/ab2_r3_vs_r4.txt
