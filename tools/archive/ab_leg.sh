#!/bin/bash
# GPU box: A/B of library builds on ONE box for any bench leg, alternating:  tools/ab_leg.sh <variant.so> <rounds> <bench flags...>
V=$1; R=$2; shift 2
for i in $(seq $R); do for lib in "" $V; do LYRA_HIP_LIB=$lib python bench.py "$@" --no-cpu-baseline --steps 400 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default}', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items() if k.startswith(('resample','logmel'))))"; done; done
