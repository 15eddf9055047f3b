for args in "--steps 20 --warmup 3" "--steps 20 --warmup 3 --no-kernel-table" "--steps 20 --warmup 30" "--steps 20 --warmup 3 --per-call" "--steps 100 --warmup 3" "--steps 20 --warmup 3 --latency-steps 0"; do
python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', r['value'], r['ms_per_step'], 'enq', r.get('host_enqueue_ms'), (r.get('dominant_kernel') or {}).get('avg_us'), (r.get('dominant_kernel') or {}).get('launches'))"
done
