#!/bin/bash
# GPU box: several library variants against the default, alternating, whole-step throughput + serial table
for i in 1 2; do for lib in "" "$@"; do LYRA_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default}', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))"; done; done
