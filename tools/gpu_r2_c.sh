#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
b() { python bench.py --no-cpu-baseline --no-kernel-table --steps 200 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("secondary"))'; }
echo "c4 nsub1: $(b --config 4)"
echo "c4 nsub2: $(LYRA_HIP_SUBBATCHES=2 b --config 4)"
echo "c3 nsub1: $(b)"
echo "c3 nsub2: $(LYRA_HIP_SUBBATCHES=2 b)"
echo "c5 nsub1: $(b --config 5)"
echo "c5 nsub2: $(LYRA_HIP_SUBBATCHES=2 b --config 5)"
echo "c2 nsub1: $(b --config 2)"
