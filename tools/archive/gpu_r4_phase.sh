cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { env LYRA_HIP_LIB=lyra_amd/variants/phase.so "$@" timeout 300 python bench.py --no-cpu-baseline --no-verify --no-kernel-table --warmup 4 --ramp-steps 0 --steps 1500 --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"; }
for i in 1 2; do
for us in 0 20 40 60 80 100 130 160 200 250; do run LYRA_HIP_PHASE_US=$us; done
for us in 40 80 130 200; do run LYRA_HIP_PHASE_US=$us LYRA_HIP_PHASE_ENC=1; done
done | tee gpurun_out/r04/phase_shift.txt
