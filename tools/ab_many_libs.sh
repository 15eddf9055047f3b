#!/bin/bash
# GPU box: sustained bare step for several library builds on ONE box, alternating:  tools/ab_many_libs.sh <rounds> lib1.so lib2.so ...
R=$1; shift
for i in $(seq $R); do for lib in "" "$@"; do LYRA_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 400 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default}', r['value'], r['ms_per_step'], ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))"; done; done
