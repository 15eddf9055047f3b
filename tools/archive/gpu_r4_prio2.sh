cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { local label=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-verify --no-kernel-table --steps 1000 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', r['value'], r['ms_per_step'])"; }
for i in 1 2 3 4 5 6; do
run default A=1
run flat LYRA_HIP_FLAT_PRIO=1
run q_high LYRA_HIP_PRIO=0,0,2
run dec_mid LYRA_HIP_PRIO=0,1,0
done | tee gpurun_out/r04/prio_ab2.txt
