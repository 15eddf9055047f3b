#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
LYRA_HIP_FUSED=12 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for f in 0 4 8 12; do
  echo "fused=$f | $(LYRA_HIP_FUSED=$f MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
  echo "fused=$f | $(LYRA_HIP_FUSED=$f python bench.py --no-cpu-baseline --no-kernel-table --steps 300 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
