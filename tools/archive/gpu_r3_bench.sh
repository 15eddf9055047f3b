#!/bin/bash
# GPU box: round-3 bench legs (default driver form first), JSON lines under gpurun_out/$1
out=gpurun_out/${1:-r03_bench}
mkdir -p $out
python bench.py > $out/config3.json 2> $out/config3.err
python bench.py --steps 20 --warmup 3 > $out/config3_driver_form.json 2>> $out/config3.err
python bench.py --full-decoder --no-cpu-baseline > $out/config3_full_decoder.json 2>> $out/config3.err
python bench.py --dtx --no-cpu-baseline > $out/config3_dtx.json 2>> $out/config3.err
python bench.py --rate 48000 --no-cpu-baseline > $out/config3_48k.json 2>> $out/config3.err
python bench.py --full-decoder --dtx --rate 48000 --no-cpu-baseline > $out/config3_full_dtx_48k.json 2>> $out/config3.err
python bench.py --per-call --no-cpu-baseline > $out/config3_per_call.json 2>> $out/config3.err
python bench.py --config 4 --no-cpu-baseline > $out/config4.json 2>> $out/config3.err
python bench.py --config 2 --no-cpu-baseline > $out/config2.json 2>> $out/config3.err
python bench.py --config 5 --no-cpu-baseline --steps 200 > $out/config5.json 2>> $out/config3.err
for f in $out/*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    lat = r.get("step_latency_us") or {}
    print(sys.argv[1].split('/')[-1], "frames/s", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"],
          "lat p50/p99/max", lat.get("p50"), lat.get("p99"), lat.get("max"),
          " ".join(f"{k.replace('_kernel','')}={v['avg_us']}" for k, v in (r.get("kernels") or {}).items()))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -5 $out/config3.err
