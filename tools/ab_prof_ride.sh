#!/bin/bash
# GPU box: the in-region brackets as event records vs riding on the kernels' own packets, driver form, alternating
for i in 1 2 3 4 5; do
for mode in ride records; do
  if [ $mode = records ]; then export LYRA_HIP_PROF_RECORDS=1; else unset LYRA_HIP_PROF_RECORDS; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=r['dominant_kernel']; o=r['dominant_kernels_also_bracketed'][0]
print('$mode', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('verified'), d['kernel'], d['avg_us'], d['launches'], o['kernel'], o['avg_us'], ' '.join('%s=%.1f'%(k.replace('_kernel',''),v['avg_us']) for k,v in r['kernels'].items()))"
done; done
