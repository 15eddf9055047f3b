for P in "0,2,0" "0,2,1" "0,2,2" "1,2,0" "0,1,2" "1,2,2" "1,1,1"; do
for K in 20 400; do
LYRA_HIP_PRIO=$P python bench.py --steps $K --warmup 3 --no-cpu-baseline --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $P K=$K', r['value'], r['ms_per_step'])"
done; done
