#!/bin/bash
# GPU box: stage pipeline (LYRA_HIP_STAGED) x CU masks x HW queues at configs 2 and 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
H="00ff00ff,ff00ff00,00ff00ff,ff00ff00"
run() { # label config env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline ${VERIFY:---no-verify} --steps 1000 --latency-steps 0 2>gpurun_out/r04/staged_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label config=$cfg', r['value'], r['ms_per_step'], 'verified=%s' % r.get('verified'), 'enq_ms=%s' % r.get('host_enqueue_ms'), ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))" || tail -5 gpurun_out/r04/staged_err.txt; }
{
VERIFY=" " run staged_masks_VERIFIED 2 LYRA_HIP_STAGED=1 LYRA_HIP_CU_MASKS=$H
VERIFY=" " run staged_VERIFIED 3 LYRA_HIP_STAGED=1
for i in 1 2; do
  run default 2 A=1
  run masks 2 LYRA_HIP_CU_MASKS=$H
  run staged 2 LYRA_HIP_STAGED=1
  run staged_masks 2 LYRA_HIP_STAGED=1 LYRA_HIP_CU_MASKS=$H
  run staged_masks_hwq8 2 LYRA_HIP_STAGED=1 LYRA_HIP_CU_MASKS=$H GPU_MAX_HW_QUEUES=8
  run staged_hwq8 2 LYRA_HIP_STAGED=1 GPU_MAX_HW_QUEUES=8
  run staged_flat_hwq8 2 LYRA_HIP_STAGED=1 GPU_MAX_HW_QUEUES=8 LYRA_HIP_FLAT_PRIO=1
done
run default 3 A=1
run staged 3 LYRA_HIP_STAGED=1
run staged_hwq8 3 LYRA_HIP_STAGED=1 GPU_MAX_HW_QUEUES=8
run staged_masks_hwq8 3 LYRA_HIP_STAGED=1 LYRA_HIP_CU_MASKS=$H GPU_MAX_HW_QUEUES=8
} 2>&1 | tee gpurun_out/r04/staged_pipeline.txt
