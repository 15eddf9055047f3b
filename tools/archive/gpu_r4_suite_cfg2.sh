#!/bin/bash
# GPU box: full GPU suite, then the verified config-2 and config-3 bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --config 2 > gpurun_out/r04/bench_config2.json 2>gpurun_out/r04/bench_config2.err; tail -c 1500 gpurun_out/r04/bench_config2.json | head -c 600; echo
timeout 600 python bench.py --config 2 --streams 512 --no-cpu-baseline --latency-steps 0 | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=512', r['value'], r['ms_per_step'], r.get('verified'))"
