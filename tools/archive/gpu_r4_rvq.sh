#!/bin/bash
# GPU box: the screened quantizer: RVQ tests first, full suite, then A/B vs the all-exact chain kernel (variant build, LYRA_HIP_RVQ_WIDE=1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q -k "rvq or parity" 2>&1 | tail -6
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
run() { local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline ${VERIFY:---no-verify} --steps 1000 --latency-steps 0 2>gpurun_out/r04/rvq_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label config=$cfg', r['value'], r['ms_per_step'], 'verified=%s' % r.get('verified'), ' '.join(k.replace('_kernel','')+'='+str(v['avg_us']) for k,v in r['kernels'].items()))" || tail -5 gpurun_out/r04/rvq_err.txt; }
P=LYRA_HIP_LIB=lyra_amd/variants/parked.so
{
VERIFY=" " run screened_VERIFIED 3 A=1
for i in 1 2 3; do
  run chain 3 $P LYRA_HIP_RVQ_WIDE=1
  run screened 3 A=1
done
run chain 2 $P LYRA_HIP_RVQ_WIDE=1
run screened 2 A=1
} 2>&1 | tee gpurun_out/r04/rvq_screened_ab.txt
