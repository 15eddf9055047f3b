#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
b() { python bench.py --no-cpu-baseline --no-kernel-table --steps 300 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for i in 1 2; do
echo "touch64: $(b)"
echo "no touch64: $(LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/nt64.so b)"
done
echo "touch64 $(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
echo "no      $(LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/nt64.so MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
