for i in 1 2 3; do for args in "--steps 20 --warmup 3" "--steps 20 --warmup 3 --no-kernel-table" "--steps 20 --warmup 3 --ramp-steps 200"; do
python bench.py $args --no-cpu-baseline --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', r['value'], r['ms_per_step'], (r.get('dominant_kernel') or {}).get('launches'))"
done; done
python bench.py --no-cpu-baseline --latency-steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=1000', r['value'], r['ms_per_step'])"
