cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { local label=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-verify --no-kernel-table --steps 1000 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', r['value'], r['ms_per_step'])"; }
for i in 1 2; do
run default A=1
run flat LYRA_HIP_FLAT_PRIO=1
run enc_high LYRA_HIP_PRIO=2,0,0
run dec_q_high LYRA_HIP_PRIO=0,2,2
run q_high LYRA_HIP_PRIO=0,0,2
run enc_dec_high_q_low LYRA_HIP_PRIO=2,2,0
done | tee gpurun_out/r04/prio_ab.txt
