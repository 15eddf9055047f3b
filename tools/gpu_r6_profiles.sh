#!/bin/bash
# GPU box, round 6: GPU suite, then everything under profiles/ (TAG=r06) in one call.
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
TAG=r06 bash tools/gpu_profile_round.sh 2>&1 | tail -40
O=gpurun_out/r06
timeout 300 python bench.py --full-decoder --dtx --rate 48000 --no-cpu-baseline > $O/r06_bench_full_dtx_48k.json 2>> $O/bench.err
timeout 300 python bench.py --requant builtin_mixed --no-cpu-baseline > $O/r06_bench_builtin_mixed.json 2>> $O/bench.err
for a in "16000 0" "48000 10"; do set -- $a; timeout 300 lyra_amd/batch_bench lyra_amd/assets 4096 $1 9200 $2 200; done > $O/r06_batch_bench.jsonl 2>> $O/bench.err
timeout 300 lyra_amd/plugin_demo --bench lyra_amd/assets 2000 120 > $O/r06_plugin_boundary_bench.txt 2>&1
for f in $O/r06_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    lat = r.get("step_latency_us") or {}
    print(sys.argv[1].split('/')[-1], "frames/s", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"], "verified", r.get("verified"),
          "lat p50/p99", lat.get("p50"), lat.get("p99"),
          " ".join(f"{k.replace('_kernel','')}={v['avg_us']}" for k, v in (r.get("kernels") or {}).items()),
          "cpu", (r.get("cpu_baseline") or {}).get("value"), (r.get("cpu_baseline_xnnpack") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cut -c1-1200 $O/r06_batch_bench.jsonl; cat $O/r06_plugin_boundary_bench.txt
bash tools/plugin_mt_bench.sh > $O/r06_plugin_mt_bench.txt 2>&1; cat $O/r06_plugin_mt_bench.txt
python tools/k20_repeat.py 15 > $O/r06_k20_repeat.txt 2>&1; cat $O/r06_k20_repeat.txt
