P=$PWD/lyra_amd/variants/parked.so
run() { python bench.py --config 2 --no-cpu-baseline --latency-steps 0 --steps 2000 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], r['ms_per_step'], r.get('verified'))"; }
for i in 1 2; do
LYRA_HIP_LIB= run default
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=0 run parked_f0
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=3 run parked_f3
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=1 run parked_f1
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=2 run parked_f2
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=12 run parked_f12
LYRA_HIP_LIB=$P LYRA_HIP_FUSED=3 LYRA_HIP_CU_MASKS=0 run parked_f3_nomask
done
