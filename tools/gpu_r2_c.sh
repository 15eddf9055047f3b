#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline --no-kernel-table --steps 300 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
echo "bench: $(b)"; echo "bench: $(b)"
echo "$(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/timing.so EXIT_AT="d0:40,41,-1" python tools/pipeline_probe.py 2>&1 | grep exit_at | sed 's/full  *wall.step *//'
