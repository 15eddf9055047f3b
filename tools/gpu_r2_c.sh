#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rvq_million.py tests/test_noise_estimator.py -m gpu -x -q 2>&1 | tail -2
for w in 0 1 0 1; do
  echo "wide=$w: $(LYRA_HIP_RVQ_WIDE=$w python bench.py --no-cpu-baseline --no-kernel-table --steps 300 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
echo "slim $(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
echo "wide $(LYRA_HIP_RVQ_WIDE=1 MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
