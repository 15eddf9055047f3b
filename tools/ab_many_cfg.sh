#!/bin/bash
# GPU box: A/B of library builds on ONE box, alternating:  tools/ab_many_cfg.sh "<bench args>" rounds lib1.so lib2.so ...
# ("default" = the shipped library)
ARGS=$1; R=$2; shift 2
for i in $(seq $R); do for lib in "$@"; do
  L=$lib; [ "$lib" = default ] && L=""
  LYRA_HIP_LIB=$L python bench.py $ARGS --no-cpu-baseline --latency-steps 0 --no-verify 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r.get('kernels',{}).items()))"
done; done
