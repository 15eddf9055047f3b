#!/bin/bash
# GPU box: the round-3 closing set in one call -- GPU tests, every bench leg, rocprofv3 trace + PMC passes, public-API bench.
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
TAG=r03 bash tools/gpu_profile_round.sh 2>&1 | tail -30
O=gpurun_out/r03
timeout 300 python bench.py --full-decoder --dtx --rate 48000 --no-cpu-baseline > $O/r03_bench_full_dtx_48k.json 2>> $O/bench.err
timeout 300 python bench.py --per-call --no-cpu-baseline > $O/r03_bench_per_call.json 2>> $O/bench.err
for a in "16000 0" "48000 10" "16000 10"; do set -- $a; timeout 300 lyra_amd/batch_bench lyra_amd/assets 4096 $1 9200 $2 200; done > $O/r03_batch_bench.jsonl 2>> $O/bench.err
for f in $O/r03_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    lat = r.get("step_latency_us") or {}
    print(sys.argv[1].split('/')[-1], "frames/s", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"],
          "lat p50/p99", lat.get("p50"), lat.get("p99"),
          " ".join(f"{k.replace('_kernel','')}={v['avg_us']}" for k, v in (r.get("kernels") or {}).items()))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cut -c1-400 $O/r03_batch_bench.jsonl
