#!/bin/bash
# GPU box: sustained throughput under environment toggles, alternating:  tools/env_ab.sh "VAR=1" ["VAR2=1" ...]
for i in 1 2 3; do for e in "X_NONE=1" "$@"; do env $e python bench.py --no-cpu-baseline --no-kernel-table --latency-steps 0 --steps 1500 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', r['value'], r['ms_per_step'])"; done; done
