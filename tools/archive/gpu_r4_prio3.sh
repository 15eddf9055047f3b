cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { local label=$1; local args=$2; shift 2; env "$@" timeout 300 python bench.py $args --no-cpu-baseline --no-verify --no-kernel-table --steps 600 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label [$args]', r['value'], r['ms_per_step'])"; }
for i in 1 2 3; do
for a in "--config 5" "--config 4" "--full-decoder" "--dtx" "--rate 48000" "--steps 20 --warmup 5"; do
run new "$a" A=1
run old_dec_high "$a" LYRA_HIP_PRIO=0,2,0
done; done | tee gpurun_out/r04/prio_ab3.txt
