#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -2
b() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("dominant_kernel",{}).get("avg_us"))'; }
echo "default (table + sampled events): $(b) $(b) $(b)"
echo "no table: $(b --no-kernel-table) $(b --no-kernel-table)"
