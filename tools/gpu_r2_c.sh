#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline --no-kernel-table --steps 300 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
echo "bench: $(b)"; echo "bench: $(b)"
echo "c2: $(b --config 2)"; echo "c5: $(b --config 5)"; echo "logmel: $(b --with-logmel)"
echo "$(MODES=full python tools/pipeline_probe.py 2>&1 | grep '^full')"
