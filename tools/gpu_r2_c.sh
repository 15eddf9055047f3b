#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2c; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
for v in base new base new; do
echo "== $v"
if [ $v = base ]; then export LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/base.so; else unset LYRA_HIP_LIB; fi
MODES=full python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/codewarm2.txt
python bench.py --steps 100 --no-cpu-baseline --no-kernel-table | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
done
