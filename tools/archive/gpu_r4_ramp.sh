cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
for i in 1 2 3 4; do for r in 64 256 1024; do timeout 300 python bench.py --steps 20 --warmup 5 --ramp-steps $r --no-cpu-baseline --no-verify --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ramp $r', r['value'], r['ms_per_step'])"; done; done | tee gpurun_out/r04/ramp_steps.txt
