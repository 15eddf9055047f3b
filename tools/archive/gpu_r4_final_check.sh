cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form', r['value'], r['ms_per_step'], r['roofline']['frac'], r['verified'])"; done
timeout 300 python bench.py > gpurun_out/r04/r04_bench_config3.json 2>/dev/null; python -c "
import json
r=json.loads(open('gpurun_out/r04/r04_bench_config3.json').read().strip().splitlines()[-1]); print('sustained', r['value'], r['ms_per_step'], r['roofline']['frac'], r['verified'], r['step_latency_us']['p50'], r['step_latency_us']['p99'])"
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/r04_bench_driver_form_k20.json 2>/dev/null
