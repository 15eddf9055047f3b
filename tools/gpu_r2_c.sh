#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2c; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 )
MODES=full python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids
python bench.py --steps 100 --no-cpu-baseline --no-kernel-table | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
