#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY'
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,"priority_range") else None)
PY
for pr in "0,0" "0,-1" "-1,0" "1,-1" "0,-2"; do
  echo "prio $pr: $(LYRA_HIP_PRIO=$pr python bench.py --no-cpu-baseline --no-kernel-table --steps 300 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
