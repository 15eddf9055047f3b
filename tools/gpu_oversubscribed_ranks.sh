#!/bin/bash
# GPU box with ONE GPU: the N-rank bench (the driver's launcher command) with every rank on GPU 0 -- hardware smoke of the
# multi-process path (processes, contexts, CPU shares, barriers, reductions, rank gather, CPU legs after the group is gone).
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4 8; do
  echo "== $N ranks"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
    bench.py --gpus $N --steps 20 --warmup 5 --oversubscribe-device 0 2> gpurun_out/oversub_$N.err | tail -1 > gpurun_out/oversub_$N.json
  python - "$N" <<'PY'
import json, sys
N = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/oversub_{N}.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms/step", r["ms_per_step"], "n_gpus", r["n_gpus"], "ranks", r.get("ranks"), "verified", r.get("verified"),
          "cpus_per_rank", r.get("cpus_per_rank"), "per-rank ms", r.get("per_rank_ms_per_step"))
    print("  ", r["config"]["parallelism"][:160])
    print("  cpu_baseline", (r.get("cpu_baseline") or {}).get("value"), (r.get("cpu_baseline") or {}).get("cores"), "xnnpack", (r.get("cpu_baseline_xnnpack") or {}).get("value"))
    print("  ", r.get("oversubscribed"))
except Exception as e:
    print("FAILED", e); print(open(f"gpurun_out/oversub_{N}.err").read()[-1500:])
PY
done
echo "== typed without a launcher (self-spawn), 2 ranks"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --oversubscribe-device 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('value', r['value'], 'ranks', r.get('ranks'), 'verified', r.get('verified'), r['config']['parallelism'][-60:])"
