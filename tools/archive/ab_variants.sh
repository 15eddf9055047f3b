#!/bin/bash
# A/B of liblyra_hip.so build variants on one box: serialised per-kernel times + overlapped bench value.
#   VARIANTS="cur A B" bash tools/ab_variants.sh      (cur = lyra_amd/liblyra_hip.so, others = lyra_amd/variants/<name>.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
for v in $VARIANTS; do
  if [ $v = cur ]; then unset LYRA_HIP_LIB; else export LYRA_HIP_LIB=$GRAFT_REPO_ROOT/lyra_amd/variants/$v.so; fi
  s=$(MODES=full python tools/pipeline_probe.py 2>&1 | grep "^full")
  b=$(python bench.py --steps 100 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$v | $s | bench $b" | tee -a $O/ab.txt
done
done
