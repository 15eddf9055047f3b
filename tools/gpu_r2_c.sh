#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-kernel-table "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for i in 1 2 3 4; do
echo "base: $(b)   int8 stages at 96 VGPRs: $(LYRA_HIP_LIB=$R/lyra_amd/variants/i8w5.so b)"
done
echo "base $(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
echo "i8w5 $(LYRA_HIP_LIB=$R/lyra_amd/variants/i8w5.so MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
