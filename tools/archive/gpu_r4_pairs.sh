#!/bin/bash
# GPU box: pairwise-fused stage kernels (parked variant, LYRA_HIP_FUSED bits 4 = enc s1+s2, 8 = dec s0+s1) at config 3 and 2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
run() { local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline ${VERIFY:---no-verify} --steps 1000 --latency-steps 0 2>gpurun_out/r04/pairs_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label config=$cfg', r['value'], r['ms_per_step'], 'verified=%s' % r.get('verified'), ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))" || tail -5 gpurun_out/r04/pairs_err.txt; }
P=LYRA_HIP_LIB=lyra_amd/variants/parked.so
{
VERIFY=" " run pairs12_VERIFIED 3 $P LYRA_HIP_FUSED=12
for i in 1 2 3; do
  run default 3 A=1
  run pair_enc 3 $P LYRA_HIP_FUSED=4
  run pair_dec 3 $P LYRA_HIP_FUSED=8
  run pairs 3 $P LYRA_HIP_FUSED=12
done
run default 2 A=1
run pairs 2 $P LYRA_HIP_FUSED=12
} 2>&1 | tee gpurun_out/r04/pair_fusion.txt
