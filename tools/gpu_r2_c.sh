#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
echo "ext events: $(b) $(b) $(b)"
echo "prev      : $(LYRA_HIP_LIB=$R/lyra_amd/variants/prev3.so b) $(LYRA_HIP_LIB=$R/lyra_amd/variants/prev3.so b)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --no-cpu-baseline --no-kernel-table --steps 200 > /tmp/tl.log 2>&1
python $R/tools/timeline.py $(find /tmp/tl -name "*.db" | head -1) 14
