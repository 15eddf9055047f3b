#!/bin/bash
# GPU box: sustained throughput with stage kernels launched as k slices (LYRA_HIP_TILE_DIV_<KERNEL>=k, api.hip tile_div):
# does bounding one chain's residency let the other chain's kernel live beside it?   tools/tile_div_probe.sh
D() { local s=""; for k in "$@"; do s="$s LYRA_HIP_TILE_DIV_${k%%=*}=${k##*=}"; done; echo $s; }
CASES=(
 "X_NONE=1"
 "$(D ENC_S0=2 ENC_S1=2 ENC_S2=2 DEC_S0=2 DEC_S1=2 DEC_S2=2)"
 "$(D ENC_S0=2)"
 "$(D ENC_S2=2 DEC_S0=2)"
 "$(D ENC_S0=2 ENC_S1=2 DEC_S1=2 DEC_S2=2)"
 "$(D ENC_S0=2 ENC_S1=2 ENC_S2=2)"
 "$(D DEC_S0=2 DEC_S1=2 DEC_S2=2)"
 "$(D ENC_S0=4 ENC_S1=2 ENC_S2=2 DEC_S0=2 DEC_S1=2 DEC_S2=4)"
)
for i in 1 2; do for e in "${CASES[@]}"; do env $e python bench.py --no-cpu-baseline --no-kernel-table --latency-steps 0 --steps 1500 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], '$e'.replace('LYRA_HIP_TILE_DIV_',''))"; done; done
