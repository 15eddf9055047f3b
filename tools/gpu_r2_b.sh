#!/bin/bash
# A/B: old block-per-stream state layout (variants/base.so) vs per-kernel regions (liblyra_hip.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $O
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/base.so; else unset LYRA_HIP_LIB; fi
  timeout 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  echo "$v $rep rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_${v}_$rep.json"))
print("$v", d["value"], d["ms_per_step"], d["serial_sum_us"], {k.replace("_kernel",""):v["avg_us"] for k,v in d["kernels"].items()})
PY
done
done
unset LYRA_HIP_LIB
( timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -3 $O/pytest.log
