#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_side_kernels.py -m gpu -x -q 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --no-kernel-table "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for i in 1 2 3; do
echo "new barriers: $(b)   old: $(LYRA_HIP_LIB=$R/lyra_amd/variants/oldbar.so b)"
done
echo "new $(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
echo "old $(LYRA_HIP_LIB=$R/lyra_amd/variants/oldbar.so MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
