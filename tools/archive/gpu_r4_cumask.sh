#!/bin/bash
# GPU box: CU-masked streams (LYRA_HIP_CU_MASKS = e,d,q,n hex patterns) at configs 2 and 3, alternating with the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
run() { # label config masks [extra env]
  env LYRA_HIP_CU_MASKS="$3" $4 timeout 300 python bench.py --config $2 --no-cpu-baseline --no-verify --steps 1000 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 config=$2 masks=$3 $4', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))"; }
for i in 1 2; do
  run default 2 ""
  run halves 2 "00ff00ff,ff00ff00,00ff00ff,ff00ff00"
  run halves_q_free 2 "00ff00ff,ff00ff00,0,0"
  run halves_q_dec 2 "00ff00ff,ff00ff00,ff00ff00,ff00ff00"
  run nomask_flatprio 2 "" LYRA_HIP_FLAT_PRIO=1
done 2>&1 | tee gpurun_out/r04/cumask_config2.txt
for i in 1 2; do
  run default 3 ""
  run halves 3 "00ff00ff,ff00ff00,00ff00ff,ff00ff00"
  run halves_q_free 3 "00ff00ff,ff00ff00,0,0"
done 2>&1 | tee gpurun_out/r04/cumask_config3.txt
