#!/bin/bash
# GPU box: fused side kernels (parked variant) x CU masks at config 2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
H="00ff00ff,ff00ff00,00ff00ff,ff00ff00"
run() { local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline ${VERIFY:---no-verify} --steps 1000 --latency-steps 0 2>gpurun_out/r04/fm_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label config=$cfg', r['value'], r['ms_per_step'], 'verified=%s' % r.get('verified'), 'enq_ms=%s' % r.get('host_enqueue_ms'), ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))" || tail -5 gpurun_out/r04/fm_err.txt; }
P=LYRA_HIP_LIB=lyra_amd/variants/parked.so
{
VERIFY=" " run fused3_masks_VERIFIED 2 $P LYRA_HIP_FUSED=3 LYRA_HIP_CU_MASKS=$H
for i in 1 2; do
  run masks 2 LYRA_HIP_CU_MASKS=$H
  run fused3_masks 2 $P LYRA_HIP_FUSED=3 LYRA_HIP_CU_MASKS=$H
  run fused1_masks 2 $P LYRA_HIP_FUSED=1 LYRA_HIP_CU_MASKS=$H
  run fused2_masks 2 $P LYRA_HIP_FUSED=2 LYRA_HIP_CU_MASKS=$H
  run fused3 2 $P LYRA_HIP_FUSED=3
done
} 2>&1 | tee gpurun_out/r04/fused_masks_config2.txt
